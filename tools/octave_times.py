#!/usr/bin/env python3
"""Per-octave duration of the pyramid kernel from a rocprofv3 --kernel-trace CSV: the dispatches of
one launch set come in octave order, so dispatch i of the kernel belongs to octave i mod n_octaves.
usage: octave_times.py <dir with *kernel_trace.csv> [n_octaves]"""
import csv
import glob
import os
import sys

root = sys.argv[1]
n_oct = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "pyramid_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
per = [[] for _ in range(n_oct)]
for i, (a, b) in enumerate(rows):
    per[i % n_oct].append((b - a) / 1e3)
for o, v in enumerate(per):
    if v:
        v = sorted(v)
        print(f"octave {o}: median {v[len(v)//2]:9.1f} us  min {v[0]:9.1f}  ({len(v)} launches)")
print("octave 0 launches in time order (us):", " ".join(f"{x:.0f}" for x in per[0]))
print(f"sum of medians: {sum(sorted(v)[len(v)//2] for v in per if v):.1f} us")
