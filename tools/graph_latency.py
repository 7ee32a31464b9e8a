#!/usr/bin/env python3
"""One image through sift_extract_noalloc, called directly and replayed as a HIP graph (the entry
allocates nothing, so it can be captured): median wall time of launch + wait, image on the device.
usage: graph_latency.py [width height]   (the image's scratch must fit the entry's static buffer)"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift_amd as sift  # noqa: E402

w = int(sys.argv[1]) if len(sys.argv) > 1 else 800
h = int(sys.argv[2]) if len(sys.argv) > 2 else 640
sift.set_params(n_scales=2, n_octaves=4, max_points=2500)
sift.initDeviceConstant()
d_img = torch.from_numpy(sift.synth_image(w, h, seed=1000)).cuda()
d_pts = torch.empty((2500, sift.POINT_FLOATS), dtype=torch.float32, device="cuda")
d_count = torch.zeros((1,), dtype=torch.int32, device="cuda")
args = (C.c_void_p(d_pts.data_ptr()), C.c_void_p(d_count.data_ptr()), C.c_void_p(d_img.data_ptr()), w, w, h)
stream = torch.cuda.Stream()
L = sift.lib()


def direct():
    assert L.sift_extract_noalloc(*args, C.c_void_p(stream.cuda_stream)) == 0, sift.last_error()
    stream.synchronize()


direct()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=stream):
    assert L.sift_extract_noalloc(*args, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0, sift.last_error()


def replay():
    graph.replay()
    torch.cuda.synchronize()


for name, fn in (("direct call + stream wait", direct), ("graph replay + device wait", replay)):
    for _ in range(20):
        fn()
    t = []
    for _ in range(200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    t = np.sort(np.array(t)) * 1e6
    print(f"{name:28s} {w}x{h}: median {t[100]:7.1f} us  p10 {t[20]:7.1f}  p90 {t[180]:7.1f}  ({int(d_count.item())} keypoints)")
