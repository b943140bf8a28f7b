#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
for e in 1e-3 1e-3 1e-3 1e-8 1e-8; do echo eps $e; python tools/graph_debug.py 24 $e 2>&1 | tail -3; done
