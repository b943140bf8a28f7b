#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}; python tools/mfma_fill_probe.py 2>&1 | tail -16
