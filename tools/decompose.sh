#!/bin/bash
# Where the fused pyramid kernel's time goes: builds libsift_amd.so variants with the write
# traffic and/or the column arithmetic compiled out (PYR_EXP_* in sift_kernels.hip) and runs
# bench.py on each.  Build step here (CPU), run step on the GPU box:
#   tools/decompose.sh build      -> sift_amd/libsift_exp_*.so
#   tools/decompose.sh run        -> table on stdout (results of the variants are wrong on purpose)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fPIC -std=c++17 -Wno-unused-result"
if [ "${1:-}" = build ]; then
  cd "$R/sift_amd/csrc" && make -s
  for v in NO_STORES NO_COLUMN_FMA "NO_STORES -DPYR_EXP_NO_COLUMN_FMA"; do
    tag=$(echo "$v" | sed 's/ -DPYR_EXP_/_/')
    /opt/rocm/bin/hipcc $F -DPYR_EXP_$v -c sift_kernels.hip -o /tmp/sk_$tag.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libsift_exp_$tag.so /tmp/sk_$tag.o sift_host.o synth.o pipeline.o -lm -lpthread
  done
  exit 0
fi
cd "$R"
line() { python3 bench.py --no-cpu-baseline --no-ceiling 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-28s %7.0f images/s   pyramid %6.2f us/image' % (sys.argv[1], d['value'], d['kernels_ms_per_image']['pyramid']*1e3))" "$1"; }
line "full kernel"
for tag in NO_COLUMN_FMA NO_STORES NO_STORES_NO_COLUMN_FMA; do
  SIFT_AMD_LIB=$R/sift_amd/libsift_exp_$tag.so line "$tag"
done
