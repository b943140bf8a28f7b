cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_bnact.py -x -q -m gpu -k "se_" 2>&1 | tail -2
for c in cfg3 cfg4; do timeout 300 python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-150; done
