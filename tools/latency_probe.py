#!/usr/bin/env python3
"""Runs the reference's call sequence (initDeviceVariables + extractSift + getPointCount) on one
resident image a number of times and prints the median wall time; meant to be run under
`rocprofv3 --kernel-trace` as well, for tools/timeline.py.
usage: latency_probe.py [--iters N] [--strip-rows R] [--size WxH] [--queue-cap C] [--photo]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift_amd as sift  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--strip-rows", type=int, default=0)
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--queue-cap", type=int, default=-1)
ap.add_argument("--photo", action="store_true", help="photograph density: the committed crop of Sift/img/img1.ppm tiled to the size")
a = ap.parse_args()
w, h = (int(v) for v in a.size.split("x"))
sift.set_params(n_octaves=5, n_scales=2, max_points=2500)
sift.initDeviceConstant()
if a.strip_rows:
    sift.lib().sift_set_strip_rows(a.strip_rows)
if a.queue_cap >= 0:
    sift.lib().sift_set_queue_cap(a.queue_cap)
img = sift.synth_image(w, h, seed=1000)
if a.photo:
    g = np.load(os.path.join(ROOT, "tests", "golden", "real_img1_crop.npz"))["gray"].astype(np.float32)
    row = np.concatenate([g, g[:, ::-1]] * (w // 640 + 1), axis=1)
    img = np.ascontiguousarray(np.concatenate([row, row[::-1]] * (h // 512 + 1), axis=0)[:h, :w])
d = torch.from_numpy(img).cuda()
pts = sift.new_point_buffer()


def one():
    sift.initDeviceVariables()
    sift.extractSift(pts, d, w, w, h, None, 0, 0)
    return sift.getPointCount(0)


for _ in range(20):
    n = one()
t = []
for _ in range(a.iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = one()
    t.append(time.perf_counter() - t0)
t = np.sort(np.array(t)) * 1e6
print(f"{w}x{h} strip_rows={a.strip_rows} queue_cap={a.queue_cap}: median {t[len(t) // 2]:7.1f} us  "
      f"p10 {t[len(t) // 10]:7.1f}  p90 {t[(9 * len(t)) // 10]:7.1f}  ({n} keypoints)")
