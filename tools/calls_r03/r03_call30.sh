#!/bin/bash
# counters of the R = 32 devoxelize gather (op level)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03zb; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_'); rm -rf /tmp/pmc_$n
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$n -- python $R/tools/opbench.py --ops devox_fwd --shapes 16x64x4096x32 --iters 5 > /dev/null 2>&1
  python $R/tools/pmc_by_kernel.py /tmp/pmc_$n gather_lds > $O/pmc_$n.txt 2>&1
done
cat $O/pmc_*.txt | cut -c1-120
