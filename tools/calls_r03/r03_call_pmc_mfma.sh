R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py"
for c in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  rm -rf /tmp/pmc_m; timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_m -- $BENCH --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/pmc_by_kernel.py /tmp/pmc_m pvcnn > $O/pmc_${n}_bench.txt
done
ls -la $O | grep pmc_SQ
