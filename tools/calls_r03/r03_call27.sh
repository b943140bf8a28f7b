#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}; for i in 1 2 3 4 5; do timeout 300 python -m pytest tests/test_gpu_optim.py tests/test_gpu_graph.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed" | cut -c1-420 | head -6; done
