#!/bin/bash
# copy the evidence set of gpurun_out/$ROUND (tools/collect_evidence.sh) into profiles/ as ${ROUND}_<file>
RD=${ROUND:-r04}; O=gpurun_out/$RD
for f in $O/*; do b=$(basename $f); case "$b" in pmc_traffic.json|kernel_durations*.json) cp $f profiles/$b;; *) cp $f profiles/${RD}_$b;; esac; done
ls profiles | head -80
