#!/bin/bash
# Runs on the GPU box: FETCH_SIZE / WRITE_SIZE of the store-pattern micro-benchmark,
# whose kernels move a known number of bytes with the pyramid kernel's access
# widths (one dword per lane).  Calibrates the counters for that width.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=$R/gpurun_out/calib
mkdir -p "$OUT"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- ./tools/ubench/store_pattern > "$OUT/fetch.log" 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- ./tools/ubench/store_pattern > "$OUT/write.log" 2>&1 &&
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
plane = 1920 * 1080 * 4 * 64 / 1e6
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    print(f"{k:60s} FETCH_SIZE {c.get('FETCH_SIZE', 0) * 1024 / 1e6:10.1f} MB  WRITE_SIZE {c.get('WRITE_SIZE', 0) * 1024 / 1e6:10.1f} MB   (one plane = {plane:.1f} MB)")
PY
