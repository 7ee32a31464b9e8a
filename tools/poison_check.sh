#!/bin/bash
# Builds the library with -DSIFT_POISON_REGION (the keypoint kernel then fills its 23x23
# gradient region with NaN before every keypoint, so an entry that is read without having been
# computed for that keypoint turns the descriptor into NaNs), runs the GPU parity suite against
# it, and restores the normal build.  Run where hipcc and a GPU are both present, or build here
# and run the tests through gpurun.
# usage: tools/poison_check.sh [build|restore]
set -eu
cd "$(dirname "$0")/../sift_amd/csrc"
FLAGS='--offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fPIC -std=c++17 -Wno-unused-result'
case "${1:-all}" in
  build)   touch sift_kernels.hip; make HIPFLAGS="$FLAGS -DSIFT_POISON_REGION" ;;
  restore) touch sift_kernels.hip; make ;;
  all)     # the normal build comes back however the tests end
           trap 'touch sift_kernels.hip; make' EXIT
           touch sift_kernels.hip; make HIPFLAGS="$FLAGS -DSIFT_POISON_REGION"
           rc=0; (cd ../.. && python -m pytest tests -x -q -m gpu) || rc=$?
           exit $rc ;;
esac
