#!/bin/bash
# copy the evidence set of gpurun_out/$ROUND (tools/collect_evidence.sh) into profiles/ as ${ROUND}_<file>
RD=${ROUND:-r03}; O=gpurun_out/$RD
for f in $O/*; do b=$(basename $f); [ "$b" = pmc_traffic.json ] && cp $f profiles/pmc_traffic.json || cp $f profiles/${RD}_$b; done
ls profiles | head -80
