#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}; python tools/models_dev.py 2>&1 | tail -8
