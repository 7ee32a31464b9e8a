#!/bin/bash
# Attribution of the keypoint kernel's LDS bank conflicts: builds libsift_amd.so variants
# (region row stride, and each conflict-prone access switched off in turn), then one
# rocprofv3 counter pass per variant over bench.py --dense (1.9 k keypoints per image).
#   tools/kp_conflicts.sh build   (CPU, here)      tools/kp_conflicts.sh run   (GPU box)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fPIC -std=c++17 -Wno-unused-result"
VARIANTS="base:-DKP_RSTRIDE=23 stride25:-DKP_RSTRIDE=25 stride27:-DKP_RSTRIDE=27 stride29:-DKP_RSTRIDE=29 stride33:-DKP_RSTRIDE=33 no_ori_scatter:-DKP_EXP_NO_ORI_SCATTER no_desc_scatter:-DKP_EXP_NO_DESC_SCATTER no_gather:-DKP_EXP_NO_GATHER"
if [ "${1:-}" = build ]; then
  cd "$R/sift_amd/csrc" && make -s
  for v in $VARIANTS; do
    tag=${v%%:*}; def=${v#*:}
    /opt/rocm/bin/hipcc $F $def -c sift_kernels.hip -o /tmp/kp_$tag.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libsift_kp_$tag.so /tmp/kp_$tag.o sift_host.o synth.o pipeline.o -lm -lpthread &
  done
  wait
  exit 0
fi
export TMPDIR=/tmp
cd "$R"
for v in $VARIANTS; do
  tag=${v%%:*}
  export SIFT_AMD_LIB=$R/sift_amd/libsift_kp_$tag.so
  rm -rf /tmp/kp_$tag
  timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d /tmp/kp_$tag -- python3 bench.py --dense --steps 2 --no-cpu-baseline --no-ceiling --no-photo > /tmp/kp_$tag.log 2>&1 || { echo "$tag: pass failed"; tail -3 /tmp/kp_$tag.log; exit 1; }
  python3 - "$tag" $(find /tmp/kp_$tag -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
tag, path = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = 0
for r in csv.DictReader(open(path)):
    if 'keypoint_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']] += float(r['Counter_Value'])
        n += r['Counter_Name'] == 'SQ_WAVE_CYCLES'
c, a = acc['SQ_LDS_BANK_CONFLICT'], acc['SQ_LDS_IDX_ACTIVE']
print(f"{tag:16s} launches {n}  LDS conflict cycles {c/n:12.0f}  LDS active cycles {a/n:12.0f}  ratio {c/a:.3f}  LDS instr {acc['SQ_INSTS_LDS']/n:10.0f}  wave cycles {acc['SQ_WAVE_CYCLES']/n:12.0f}")
PY
done
