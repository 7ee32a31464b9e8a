#!/bin/bash
# Runs on the GPU box: one-image latency and the bench line for experiment builds of the library.
# usage: lat4.sh <tag>...
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/lat4
mkdir -p $O
cd $GRAFT_REPO_ROOT
: > $O/lat.txt
echo "normal build" >> $O/lat.txt
timeout -k 10 60 python3 tools/latency_probe.py >> $O/lat.txt 2>&1 || exit 1
for tag in "$@"; do
  echo "$tag" >> $O/lat.txt
  SIFT_AMD_LIB=$GRAFT_REPO_ROOT/sift_amd/libsift_exp_$tag.so timeout -k 10 60 python3 tools/latency_probe.py >> $O/lat.txt 2>&1 || exit 1
  SIFT_AMD_LIB=$GRAFT_REPO_ROOT/sift_amd/libsift_exp_$tag.so timeout -k 10 60 python3 tools/latency_probe.py --size 800x640 >> $O/lat.txt 2>&1 || exit 1
done
bash tools/variants.sh run "$@" > $O/variants.txt 2>&1
grep -v amdgpu.ids $O/lat.txt; cat $O/variants.txt
