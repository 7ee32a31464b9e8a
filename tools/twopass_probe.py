#!/usr/bin/env python3
"""Two-pass blur (row_blur_kernel + col_blur_dog_kernel through blurOctave with a pitch that is no
multiple of 64) on one image of each octave size of a 1080p pyramid, for several strip heights of the
column pass; run under rocprofv3 --kernel-trace, the trace holds the durations.
usage: twopass_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift_amd as sift  # noqa: E402

sift.set_params(n_octaves=5, n_scales=2, max_points=2500)
sift.initDeviceConstant()
for rows in (1, 2, 4, 8, 16, 32):
    sift.lib().sift_set_strip_rows(rows)
    for (w, h) in ((1920, 1080), (960, 540), (480, 270), (240, 135), (120, 68)):
        p = w + 4
        src = torch.rand((h, p), device="cuda") * 255
        for _ in range(4):
            out = sift.blurOctave(src, w)
        torch.cuda.synchronize()
print("done")
