#!/bin/bash
# Runs on the GPU box: device timelines of the one-image call sequence for several strip heights.
set -e
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/lat1
mkdir -p $O
cd $GRAFT_REPO_ROOT
for r in 0 1 2 4 8 16 32 64; do
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/tr$r -- python3 tools/latency_probe.py --iters 30 --strip-rows $r > $O/prof$r.log 2>&1
  python3 tools/timeline.py $O/tr$r 11 > $O/timeline$r.txt 2>&1
  rm -rf $O/tr$r
done
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/trq -- python3 tools/latency_probe.py --iters 30 --queue-cap 0 > $O/profq.log 2>&1
python3 tools/timeline.py $O/trq 11 > $O/timelineq.txt 2>&1
rm -rf $O/trq
