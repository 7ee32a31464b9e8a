#!/bin/bash
set -e
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/lat2
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 tools/twopass_probe.py > $O/prof.log 2>&1
python3 tools/trace_list.py $O/tr blur > $O/list.txt 2>&1
rm -rf $O/tr
