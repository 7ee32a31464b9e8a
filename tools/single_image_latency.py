#!/usr/bin/env python3
"""Latency of one image through the reference's entry (extractSift + getPointCount, image already
on the device) and through the batch entry with one image: median of 200 calls after 20 warm-up
calls.  usage: single_image_latency.py [width height]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift_amd as sift  # noqa: E402

w = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
sift.set_params(n_octaves=5, n_scales=2, max_points=2500)
sift.initDeviceConstant()
img = sift.synth_image(w, h, seed=1000)
d = torch.from_numpy(img).cuda()
pts = sift.new_point_buffer()


def one_extract():
    sift.initDeviceVariables()
    sift.extractSift(pts, d, w, w, h, None, 0, 0)
    return sift.getPointCount(0)


def one_batch():
    _, c = sift.extract_batch(d[None], w, d_pts=pts)
    return int(c[0])


for name, fn in (("extractSift + getPointCount", one_extract), ("sift_extract_batch, n = 1", one_batch)):
    for _ in range(20):
        n = fn()
    t = []
    for _ in range(200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = fn()
        t.append(time.perf_counter() - t0)
    t = np.sort(np.array(t)) * 1e6
    print(f"{name:30s} {w}x{h}: median {t[100]:7.1f} us  p10 {t[20]:7.1f}  p90 {t[180]:7.1f}  ({n} keypoints)")
