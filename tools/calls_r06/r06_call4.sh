#!/bin/bash
# round 6, call 4: counters (clock, MFMA pipe busy, stalls) of the wide and of the 128-row 1x1 GEMM on the same box
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
S2="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
for fill in "" "--relu"; do
for w in 1 0; do
  tag=wide${w}${fill/--/_}
  for set in 1 2; do
    if [ $set = 1 ]; then C="$S1"; else C="$S2"; fi
    D=$(mktemp -d /tmp/pmc.XXXX)
    PVCNN_PW_WIDE=$w timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python $R/tools/pw_only.py 16x1472x512x4096 $fill > $D/run.log 2>&1 || tail -3 $D/run.log
    python $R/tools/pmc_by_kernel.py $D pw_gemm --json --by-grid > $O/pmc_${tag}_set$set.json
  done
  echo "== $tag"; python $R/tools/pmc_mfma_table.py $O/pmc_${tag}_set1.json $O/pmc_${tag}_set2.json | tee $O/pmc_${tag}_table.md | cut -c1-330
done
done
