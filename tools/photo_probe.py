#!/usr/bin/env python3
"""A batch of photograph-density 1080p images (the committed crop of Sift/img/img1.ppm tiled) through
the batch entry a few times at the reference's cap of 2500; meant for rocprofv3 counter or trace
passes over the dense detector and the keypoint kernel.  usage: photo_probe.py [images] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sift_amd as sift  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = np.load(os.path.join(ROOT, "tests", "golden", "real_img1_crop.npz"))["gray"].astype(np.float32)
row = np.concatenate([g, g[:, ::-1]] * 3, axis=1)
img = np.ascontiguousarray(np.concatenate([row, row[::-1]] * 3, axis=0)[:1080, :1920])
batch = torch.from_numpy(np.stack([np.roll(img, 7 * i, axis=1) for i in range(n)])).cuda()
sift.set_params(n_scales=2, n_octaves=5, max_points=2500)
d_pts = torch.empty((n * 2500, sift.POINT_FLOATS), dtype=torch.float32, device="cuda")
sift.extract_batch(batch, 1920, d_pts=d_pts)
sift.timing_reset()
sift.timing_enable(True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    _, counts = sift.extract_batch(batch, 1920, d_pts=d_pts)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
sift.timing_enable(False)
k = {name: round(sift.timing_read(kid)[0] / (steps * n) * 1e3, 2) for name, kid in
     (("pyramid", sift.K_PYRAMID), ("verify", sift.K_VERIFY), ("detect", sift.K_DETECT), ("keypoint", sift.K_KEYPOINT))}
print(f"{n} images: {steps * n / dt:.0f} images/s, {counts.mean():.0f} keypoints per image, us per image: {k}")
# one image of the batch alone through extractSift with a large cap: what the groups hold
sift.set_params(n_scales=2, n_octaves=5, max_points=60000)
sift.lib().sift_set_dense_route_pixels(0)
pts = sift.new_point_buffer()
sift.initDeviceVariables()
sift.extractSift(pts, batch[0], 1920, 1920, 1080, None, 0, 0)
print("uncapped:", sift.getPointCount(0), "keypoints; per group (accepted, queued):", sift.group_counts(0))
