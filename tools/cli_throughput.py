#!/usr/bin/env python3
"""Images/s of the list-file front end (sift_cli: decode PGM, upload, extract, read back, write
.sift.txt) on synthetic 1080p images: the threaded list pipeline (sift_extract_list_multi) and the
reference's pairwise loop (sift), timed inside the process over a second pass of the list (the first
pass pays the GPU set-up), and the two sets of output files compared byte for byte.
usage: cli_throughput.py [n_images] [gpus]"""
import filecmp
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift_amd as sift  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
gpus = sys.argv[2] if len(sys.argv) > 2 else "1"
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
    paths = []
    for i in range(n):
        img = sift.synth_image(1920, 1080, seed=100 + i).astype(np.uint8)
        p = os.path.join(d, f"img{i:04d}.pgm")
        with open(p, "wb") as f:
            f.write(b"P5\n1920 1080\n255\n" + img.tobytes())
        paths.append(p)
    lst = os.path.join(d, "list.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    cli = os.path.join(ROOT, "sift_amd", "sift_cli")
    outs = {}
    for name, extra in (("threaded", ["--gpus", gpus]), ("pairwise", ["--pairwise"])):
        out = os.path.join(d, "out_" + name)
        os.mkdir(out)
        outs[name] = out
        r = subprocess.run([cli, lst, out, "--octaves", "5", "--timing", "--repeat", "3"] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        print(name + ":", " | ".join(line for line in r.stderr.splitlines() if line.startswith("pass")))
    names = sorted(os.listdir(outs["threaded"]))
    assert names == sorted(os.listdir(outs["pairwise"])) and len(names) == n
    same = all(filecmp.cmp(os.path.join(outs["threaded"], f), os.path.join(outs["pairwise"], f), shallow=False) for f in names)
    size = sum(os.path.getsize(os.path.join(outs["threaded"], f)) for f in names)
    print(f"output files of the two pipelines identical: {same}; text written: {size / n / 1e3:.0f} kB per image")
    assert same
