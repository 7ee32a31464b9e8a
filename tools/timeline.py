#!/usr/bin/env python3
"""Device timeline of the last launch set in a rocprofv3 --kernel-trace CSV: every dispatch with its
start relative to the set's first kernel, its duration and the idle gap in front of it.
usage: timeline.py <dir with *kernel_trace.csv> [dispatches to show from the end]"""
import csv
import glob
import os
import sys

root = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[-last:]
t0, prev = rows[0][0], rows[0][0]
busy = 0
for a, b, name in rows:
    short = name.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")
    if "anonymous" in name and "keypoint" in name:
        short = "keypoint_kernel"
    print(f"{(a - t0) / 1e3:10.1f} us  +{(b - a) / 1e3:9.1f} us  gap {(a - prev) / 1e3:7.1f} us  {short[:50]}")
    busy += b - a
    prev = b
print(f"span {(prev - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(prev - t0 - busy) / 1e3:.1f} us")
