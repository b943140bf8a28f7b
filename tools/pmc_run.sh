#!/bin/bash
# usage: tools/pmc_run.sh "<space separated counters>" <kernel-name filter> <command...>
# one rocprofv3 --pmc pass (kernel trace only, no other trace domains), per-kernel averages on stdout
R=${GRAFT_REPO_ROOT:-$PWD}; C="$1"; F="$2"; shift 2
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d /tmp/pmc.XXXX)
timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- "$@" > $D/run.log 2>&1 || tail -5 $D/run.log
python $R/tools/pmc_by_kernel.py $D "$F"
