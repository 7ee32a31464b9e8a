#!/bin/bash
# Runs bench.py on builds of the library with experiment switches (sift_amd/libsift_exp_<tag>.so,
# made here on the CPU with tools/variants.sh build <tag> "<-D flags>") and prints one line each:
# images/s and the pyramid pass's event time per image.  The normal build comes first as the base line.
# usage (GPU box): tools/variants.sh run <tag>...
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fPIC -std=c++17 -Wno-unused-result"
if [ "${1:-}" = build ]; then
  tag=$2; shift 2
  cd "$R/sift_amd/csrc" && make -s &&
  /opt/rocm/bin/hipcc $F "$@" -c sift_kernels.hip -o /tmp/sk_$tag.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libsift_exp_$tag.so /tmp/sk_$tag.o sift_host.o synth.o pipeline.o -lm -lpthread
  exit $?
fi
shift
cd "$R"
line() { timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-ceiling --no-photo ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-28s %7.0f images/s   pyramid %6.2f us/image   frac %.3f' % (sys.argv[1], d['value'], d['kernels_ms_per_image']['pyramid']*1e3, d['roofline']['frac']))" "$1"; }
line "normal build"
for tag in "$@"; do
  SIFT_AMD_LIB=$R/sift_amd/libsift_exp_$tag.so line "$tag"
done
line "normal build (again)"
