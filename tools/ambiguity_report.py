#!/usr/bin/env python3
"""Counts, over BASELINE's synthetic images, the threshold decisions of the reference's detector
(Sift/src/excludes/cudaSiftD.cu:288-293) that sit within 1e-3 of their limit or change under
roundings a CUDA compiler may have chosen.  CPU only (the oracle).  usage: ambiguity_report.py [n_images]"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import sift_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, dense in (("config 3 images (seeds 1000..)", False), ("dense images (seeds 9000..)", True)):
    pr = oracle.params(n_octaves=5, n_scales=2, max_points=1 << 20)

    def one(i):
        img = sift_amd.synth_image(1920, 1080, seed=(9000 if dense else 1000) + i, dense=dense)
        return oracle.ambiguity_report(pr, img, 1920)

    with ThreadPoolExecutor(8) as ex:
        reps = list(ex.map(one, range(n if not dense else max(1, n // 8))))
    tot = {k: sum(r[k] for r in reps) for k in reps[0]}
    print(f"{name}, {len(reps)} images of 1920x1080, 2 scales, 5 octaves:", tot)
