#!/usr/bin/env python3
"""Throughput on images with the keypoint density of a photograph: the committed 320x256 crop of
the reference's Sift/img/img1.ppm, mirrored and tiled to 1920x1080 (cap 2500 as in the reference,
and cap 20000)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sift_amd as sift  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "real_img1_crop.npz"))["gray"].astype(np.float32)
row = np.concatenate([g, g[:, ::-1]] * 3, axis=1)
img = np.concatenate([row, row[::-1]] * 3, axis=0)[:1080, :1920]
img = np.ascontiguousarray(img)
n = 64
batch = torch.from_numpy(np.stack([np.roll(img, 7 * i, axis=1) for i in range(n)])).cuda()
qcaps = [int(a) for a in sys.argv[1:]] or [65536]
for cap, qcap in [(c, q) for q in qcaps for c in (2500, 20000)]:
    sift.lib().sift_set_queue_cap(qcap)
    sift.set_params(n_scales=2, n_octaves=5, max_points=cap)
    d_pts = torch.empty((n * cap, sift.POINT_FLOATS), dtype=torch.float32, device="cuda")
    sift.extract_batch(batch, 1920, d_pts=d_pts)
    sift.timing_reset()
    sift.timing_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        _, counts = sift.extract_batch(batch, 1920, d_pts=d_pts)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sift.timing_enable(False)
    k = {name: sift.timing_read(kid)[0] / (3 * n) * 1e3 for name, kid in
         (("pyramid", sift.K_PYRAMID), ("alias", sift.K_ALIAS), ("verify", sift.K_VERIFY),
          ("detect", sift.K_DETECT), ("keypoint", sift.K_KEYPOINT))}
    print(f"queue {qcap} cap {cap}: {3 * n / dt:.0f} images/s, {counts.mean():.0f} keypoints per image, us per image:",
          {a: round(b, 2) for a, b in k.items()})
