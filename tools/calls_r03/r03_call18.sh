#!/bin/bash
# where the Conv3d igemm / wgrad kernels spend their time: SQ wait / issue counters, MFMA busy, LDS, L2
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU"; do
  n=$(echo $c | tr ' ' '_'); rm -rf /tmp/pmc_$n
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$n -- python $R/tools/convcheck.py --time --no-check --shapes 16x64x64x16,16x128x128x16,16x64x64x32 > /dev/null 2>&1
  python $R/tools/pmc_by_kernel.py /tmp/pmc_$n conv3d_ > $O/pmc_$n.txt 2>&1
done
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/tools/convcheck.py --time --no-check --shapes 16x64x64x16,16x128x128x16,16x64x64x32 > $O/convcheck.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -30 $f > $O/kernel_stats.csv
cat $O/pmc_*.txt | cut -c1-110 | head -150
