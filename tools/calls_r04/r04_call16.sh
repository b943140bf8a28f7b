#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04o; mkdir -p $O
cd $R
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --single-rank-collectives > $O/coll.out 2>$O/coll.err
tail -c 1500 $O/coll.out | cut -c1-300
