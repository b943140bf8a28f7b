cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "argmax" 2>&1 | tail -2
timeout 120 python tools/poolbench.py
