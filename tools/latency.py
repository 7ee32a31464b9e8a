#!/usr/bin/env python3
"""Single-image latency of extractSift + getPointCount (the reference's call pattern,
Sift/src/sift.cu:109-125) for the 640x480, 3 scales, 4 octaves configuration, and the
throughput of the same calls issued round-robin over the 8 stream slots."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import sift_amd as sift  # noqa: E402

w, h, S, O = 640, 480, 3, 4
sift.set_params(n_scales=S, n_octaves=O, max_points=2500)
img = torch.from_numpy(sift.synth_image(w, h, seed=1)).cuda()
pts = [sift.new_point_buffer() for _ in range(8)]
streams = [torch.cuda.Stream() for _ in range(8)]
sift.initDeviceConstant()


def one(slot):
    sift.extractSift(pts[slot], img, w, w, h, streams[slot], slot, 0)


for _ in range(5):
    sift.initDeviceVariables()
    one(0)
    n = sift.getPointCount(0)
N = 200
t0 = time.perf_counter()
for _ in range(N):
    sift.initDeviceVariables()
    one(0)
    n = sift.getPointCount(0)
dt = time.perf_counter() - t0
print(f"single slot: {dt / N * 1e6:.1f} us per image ({n} keypoints), {N / dt:.0f} images/s")
for k in (2, 4, 8):
    sift.initDeviceVariables()
    t0 = time.perf_counter()
    for _ in range(N // k):
        sift.initDeviceVariables()
        for s in range(k):
            one(s)
        for s in range(k):
            sift.getPointCount(s)
    dt = time.perf_counter() - t0
    print(f"{k} slots in flight: {(N // k) * k / dt:.0f} images/s")
sift.timing_reset()
sift.timing_enable(True)
sift.initDeviceVariables()
one(0)
sift.getPointCount(0)
sift.timing_enable(False)
for name, kid in (("pyramid", sift.K_PYRAMID), ("alias", sift.K_ALIAS), ("verify", sift.K_VERIFY),
                  ("detect", sift.K_DETECT), ("keypoint", sift.K_KEYPOINT)):
    ms, cnt = sift.timing_read(kid)
    print(f"  {name:9s} {ms * 1e3:8.1f} us in {cnt} launches")
