#!/usr/bin/env python3
"""Images/s of the list-file front end (sift_cli: load PGM, upload, extract, read back, write
.sift.txt) on synthetic 1080p images, with the time spent per stage as the library reports it."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift_amd as sift  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
with tempfile.TemporaryDirectory() as d:
    paths = []
    for i in range(n):
        img = sift.synth_image(1920, 1080, seed=100 + i).astype(np.uint8)
        p = os.path.join(d, f"img{i:03d}.pgm")
        with open(p, "wb") as f:
            f.write(b"P5\n1920 1080\n255\n" + img.tobytes())
        paths.append(p)
    lst = os.path.join(d, "list.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    out = os.path.join(d, "out")
    os.mkdir(out)
    cli = os.path.join(ROOT, "sift_amd", "sift_cli")
    for rep in range(2):
        t0 = time.perf_counter()
        r = subprocess.run([cli, lst, out, "--octaves", "5"], capture_output=True, text=True)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr
        print(f"run {rep}: {n / dt:.1f} images/s ({dt / n * 1e3:.1f} ms per image)")
    size = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
    print(f"text written: {size / n / 1e3:.0f} kB per image")
