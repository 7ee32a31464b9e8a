#!/usr/bin/env python3
"""Lists the dispatches of a rocprofv3 --kernel-trace CSV in start order: duration, grid, kernel.
usage: trace_list.py <dir> [substring of the kernel name]"""
import csv
import glob
import os
import sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", "")))
rows.sort()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for a, b, name, gx, gy, gz in rows:
    if pat and pat not in name:
        continue
    short = name.replace("void ", "").replace("(anonymous namespace)::", "")[:44]
    print(f"{(b - a) / 1e3:9.1f} us  grid {gx:>7} {gy:>5} {gz:>4}  {short}")
