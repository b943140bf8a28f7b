#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}; python tools/zero_test.py 2>&1 | tail -16
