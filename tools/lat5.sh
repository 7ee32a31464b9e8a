#!/bin/bash
# Runs on the GPU box: device timeline of the one-image call for experiment builds.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/lat5
mkdir -p $O
cd $GRAFT_REPO_ROOT
for tag in "$@"; do
  SIFT_AMD_LIB=$GRAFT_REPO_ROOT/sift_amd/libsift_exp_$tag.so timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$tag -- python3 tools/latency_probe.py --iters 30 > $O/prof_$tag.log 2>&1 || exit 1
  echo "== $tag" >> $O/timelines.txt
  python3 tools/timeline.py $O/tr_$tag 9 >> $O/timelines.txt 2>&1
  rm -rf $O/tr_$tag
done
cat $O/timelines.txt
