#!/bin/bash
# round 6, call 3: phase clocks of the wide 1x1 GEMM
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06c; mkdir -p $O
cd $R
timeout 300 python tools/phase_probe.py pw --shape 16x1472x512x4096 > $O/phase_pw_fwd.json 2> $O/phase_pw_fwd.txt; cat $O/phase_pw_fwd.txt; cut -c1-600 $O/phase_pw_fwd.json
timeout 300 python tools/phase_probe.py pw --shape 16x1472x512x4096 --bwd-data > $O/phase_pw_bwd.json 2> $O/phase_pw_bwd.txt; cat $O/phase_pw_bwd.txt; cut -c1-400 $O/phase_pw_bwd.json
timeout 300 python tools/phase_probe.py pw --shape 16x128x1024x4096 > $O/phase_pw_128.json 2> $O/phase_pw_128.txt; cat $O/phase_pw_128.txt; cut -c1-400 $O/phase_pw_128.json
timeout 600 python -m pytest tests/test_gpu_reference_composition.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
