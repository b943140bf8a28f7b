cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_conv3d.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/convcheck.py --time --no-check --shapes 8x32x32x32,8x9x32x32 2>/dev/null | grep "time_split" | grep '"nsplit": 2' | cut -c1-140
timeout 300 python bench.py --config cfg3 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-140
