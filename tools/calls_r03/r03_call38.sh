#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}; for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | cut -c1-250; done
