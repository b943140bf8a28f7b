#!/bin/bash
# round 6, call 11: the whole GPU suite + smoke + the QUICK evidence set (bench lines, traces, in-graph durations, other configs) mid-round
ROUND=r06 QUICK=1 bash tools/evidence_call.sh
