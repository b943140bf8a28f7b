set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu -k "fps or golden" 2>&1 | tail -5
timeout 300 python tools/fpsbench.py 2>&1 | grep BNM
timeout 300 python bench.py --config cfg3 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
