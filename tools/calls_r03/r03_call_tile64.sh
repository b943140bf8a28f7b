set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python tools/convcheck.py --time --no-check --shapes 8x128x128x8,8x256x256x8,8x64x128x8 2>/dev/null | grep "time_split" | cut -c1-200
timeout 300 python bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
