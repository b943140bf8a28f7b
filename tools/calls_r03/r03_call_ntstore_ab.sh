cd $GRAFT_REPO_ROOT
for v in 0 40000 150000 0 40000 150000; do
  echo "== plain<=${v}KiB"; PVCNN_AB_PLAIN_KB=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
done
for v in 0 40000; do
  echo "== cfg3 plain<=${v}KiB"; PVCNN_AB_PLAIN_KB=$v timeout 300 python bench.py --config cfg3 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
done
