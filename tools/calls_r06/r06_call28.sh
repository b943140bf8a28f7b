#!/bin/bash
# round 6, call 28: board power and shader clock (sysfs) while the matrix kernels run back to back -- are they at the power cap?
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06pw; mkdir -p $O
cd $R
ls /sys/class/drm/ > $O/drm.txt 2>&1; ls /sys/class/drm/card*/device/hwmon/hwmon*/ >> $O/drm.txt 2>&1
timeout 300 python tools/power_probe.py > $O/power_probe.jsonl 2> $O/power_probe.err; cat $O/power_probe.jsonl; tail -3 $O/power_probe.err
(rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40) > $O/rocm_smi.txt; head -30 $O/rocm_smi.txt
