#!/bin/bash
# HBM traffic and L2 hit rate of the 1x1 backward-weight and forward kernels at the classifier shape
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_'); rm -rf /tmp/pmc_$n
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$n -- python $R/tools/pwbench.py --shapes 16x1472x512x4096 > /dev/null 2>&1
  python $R/tools/pmc_by_kernel.py /tmp/pmc_$n pw_ > $O/pmc_$n.txt 2>&1
done
cat $O/pmc_*.txt | grep -A12 "pw_wgrad_f16_wide\|pw_gemm_f16_pipe" | cut -c1-120
