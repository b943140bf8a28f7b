cd $GRAFT_REPO_ROOT
for v in new old new old; do
  if [ $v = old ]; then export PVCNN_R8_OLD=1; else unset PVCNN_R8_OLD; fi
  echo "== $v"
  timeout 300 python tools/convcheck.py --time --no-check --shapes 8x128x128x8,8x256x256x8,8x64x128x8 2>/dev/null | grep "time_split" | grep '"nsplit": 2' | cut -c1-120
done
unset PVCNN_R8_OLD
for v in new old; do
  if [ $v = old ]; then export PVCNN_R8_OLD=1; else unset PVCNN_R8_OLD; fi
  timeout 300 python bench.py --config cfg3 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-140
done
