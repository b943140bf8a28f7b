#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}; python tools/det_debug2.py 2>&1 | tail -30
