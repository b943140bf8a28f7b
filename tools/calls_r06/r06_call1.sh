#!/bin/bash
# round 6, call 1: the new parity rows (returned tensors per element, adopt() of the other reference compositions, B=1 slot)
# + the baseline bench of the round-5 kernels on THIS box
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06a; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_parity_as_benched.py -m gpu -q -s -p no:cacheprovider > $O/train_parity.log 2>&1
grep "passed\|failed\|output <" $O/train_parity.log | tail -40
timeout 900 python -m pytest tests/test_gpu_reference_composition.py tests/test_gpu_concat.py -m gpu -q -s -p no:cacheprovider > $O/refcomp.log 2>&1
grep "passed\|failed\|\[adopt\]\|\[reference" $O/refcomp.log | tail -20
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
