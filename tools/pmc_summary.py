#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc CSV output per kernel: mean counter value per
dispatch, plus the gfx950 corrections for HBM traffic (FETCH_SIZE counts 128-byte
requests as 64 bytes for wide coalesced reads, so the read side is doubled;
FETCH_SIZE and WRITE_SIZE are in KiB)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for k in ("pyramid", "row_blur", "col_blur_dog", "alias_prefilter", "verify", "detect", "keypoint", "finalize",
              "fillBuffer", "copyBuffer", "subtract", "hessian", "gradient"):
        if k in name:
            return k
    return name[:40]


def main(root):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    dur = {}
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Name"])  # template instantiations of one kernel are pooled
            ns, calls = dur.get(k, (0.0, 0))
            n = int(row["Calls"])
            dur[k] = ((ns * calls + float(row["AverageNs"]) * n) / (calls + n), calls + n)
    for k in sorted(acc):
        print(f"== {k}")
        c = {n: v[0] / max(v[1], 1) for n, v in acc[k].items()}
        for n in sorted(c):
            print(f"   {n:24s} {c[n]:16.1f}  (mean of {acc[k][n][1]} dispatches)")
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            rd = c.get("FETCH_SIZE", 0) * 1024 * 2
            wr = c.get("WRITE_SIZE", 0) * 1024
            print(f"   HBM-side bytes per launch: read {rd/1e6:.2f} MB (FETCH_SIZE x2), write {wr/1e6:.2f} MB")
            if k in dur:
                ns = dur[k][0]
                print(f"   avg duration {ns/1e3:.1f} us over {dur[k][1]} calls -> {(rd+wr)/ns:.1f} GB/s counted traffic")
        if "TCC_HIT_sum" in c:
            t = c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0)
            if t:
                print(f"   L2 hit rate {c['TCC_HIT_sum']/t:.3f}")
        if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
            print(f"   LDS bank-conflict cycles / LDS active cycles = {c.get('SQ_LDS_BANK_CONFLICT',0)/c['SQ_LDS_IDX_ACTIVE']:.4f}")


    # HBM-side bytes per launch of the pyramid kernel, in the form bench.py reads
    if len(sys.argv) > 3 and "pyramid" in acc and "FETCH_SIZE" in acc["pyramid"] and "WRITE_SIZE" in acc["pyramid"]:
        import json

        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from bench import source_hash  # the counted bytes belong to these kernel sources

        c = {n: v[0] / max(v[1], 1) for n, v in acc["pyramid"].items()}
        rd, wr = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024
        json.dump({"pyramid_kernel": {"batch": int(sys.argv[3]), "hbm_bytes_per_launch": rd + wr,
                                      "read_bytes": rd, "write_bytes": wr,
                                      "source_hash": source_hash(),
                                      "source": "rocprofv3 --pmc FETCH_SIZE (x2, calibrated on dword-per-lane reads of a "
                                                "known size) and WRITE_SIZE, separate passes, mean over launches"}},
                  open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
