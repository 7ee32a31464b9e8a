#!/usr/bin/env python3
"""Randomised GPU-against-oracle comparison: sizes, scales, octaves, thresholds, profiles,
queue capacities, caps, image kinds and entry points (single image, batch, 8-bit batch) drawn at
random; every case must give the oracle's points
bit for bit (or, over the cap, a subset of them of the cap's size).
usage: tools/fuzz_parity.py [cases] [seed] [max width] [max height]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import oracle  # noqa: E402
import sift_amd as sift  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
MAX_W = int(sys.argv[3]) if len(sys.argv) > 3 else 1300
MAX_H = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
bad = 0
t0 = time.time()
for case in range(cases):
    w, h = int(rng.integers(20, MAX_W)), int(rng.integers(20, MAX_H))
    S = int(rng.integers(2, 4))
    O = int(rng.integers(1, 6))
    profile = int(rng.integers(0, 2))
    up = int(profile == 1 and rng.integers(0, 3) == 0 and O < 5)
    thr = float(rng.choice([1.0, 4.0, 10.0, 20.0]))
    rth = float(rng.choice([5.0, 10.0, 15.0]))
    qcap = int(rng.choice([0, 16, 300, 4096, 65536]))
    cap = int(rng.choice([150, 3000, 60000]))
    kind = int(rng.integers(0, 4))
    seed = int(rng.integers(0, 1 << 30))
    if kind == 0:
        img = sift.synth_image(w, h, seed=seed, dense=False)
    elif kind == 1:
        img = sift.synth_image(w, h, seed=seed, dense=True)
    elif kind == 2:
        img = np.random.default_rng(seed).integers(0, 256, (h, w)).astype(np.float32)
    else:  # smooth blobs plus steps
        yy, xx = np.mgrid[0:h, 0:w]
        img = (120 + 80 * np.sin(xx / 9.0) * np.cos(yy / 13.0) + 60 * ((xx // 37 + yy // 29) % 2)).astype(np.float32)
    pitch = w + int(rng.integers(0, 9))
    buf = np.zeros((h, pitch), np.float32)
    buf[:, :w] = img
    kw = dict(n_scales=S, n_octaves=O, max_points=cap, extrema_thresh=thr, r_thresh=rth, profile=profile, upsample=up)
    desc = dict(case=case, w=w, h=h, pitch=pitch, qcap=qcap, kind=kind, seed=seed, **kw)
    try:
        sift.set_params(**kw)
        pr = oracle.params(**kw)
        ref, _, total = oracle.extract(pr, buf, w)
        sift.lib().sift_set_queue_cap(qcap)
        # launch sets up to this many pixels skip the queues: never, always, or around this job's own size
        dense_px = int(rng.choice([0, 0, 2200000, w * h, 3 * w * h]))
        desc["dense_px"] = dense_px
        sift.lib().sift_set_dense_route_pixels(dense_px)
        route = int(rng.integers(0, 3))  # the single-image entry, the batch entry, the 8-bit batch entry
        if route == 2 and kind == 3:
            route = 1  # the 8-bit entry needs integer pixels
        desc["route"] = route
        if route == 0:
            d = torch.from_numpy(buf).cuda()
            pts = sift.new_point_buffer()
            sift.initDeviceVariables()
            sift.extractSift(pts, d, w, pitch, h, None, 0, 0)
            n = sift.getPointCount(0)
            got = sift.points_to_numpy(pts, n).view(oracle.POINT_DTYPE)
        else:
            # the image under test sits at a random place among copies of another image
            nb = int(rng.integers(1, 7))
            at = int(rng.integers(0, nb))
            chunk = int(rng.choice([1, 2, 256]))  # launch sets of 1, 2 or all images, on two streams
            desc["nb"], desc["at"], desc["chunk"] = nb, at, chunk
            sift.lib().sift_set_batch_chunk(chunk)
            other = np.roll(buf, 5, axis=0)
            stack = np.stack([buf if k == at else other for k in range(nb)])
            if route == 2:
                stack = stack.astype(np.uint8)
            pts, counts = sift.extract_batch(torch.from_numpy(stack).cuda(), w)
            n = int(counts[at])
            got = sift.points_to_numpy(pts, n, offset=at * cap).view(oracle.POINT_DTYPE)
        if total <= cap:
            ok = len(got) == len(ref) and oracle.sort_points(got).tobytes() == oracle.sort_points(ref).tobytes()
        else:
            every, _, _ = oracle.extract(oracle.params(**dict(kw, max_points=400000)), buf, w)
            have = {r.tobytes() for r in every}
            ok = len(got) == cap and all(r.tobytes() in have for r in got)
        if not ok:
            bad += 1
            print("MISMATCH", desc, "gpu", len(got), "oracle", len(ref), "total", total, flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("ERROR", desc, repr(e), flush=True)
    if case % 20 == 19:
        print(f"{case + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
sift.lib().sift_set_queue_cap(65536)
sift.lib().sift_set_dense_route_pixels(2200000)
sift.lib().sift_set_batch_chunk(256)
print("done:", cases, "cases,", bad, "bad")
sys.exit(1 if bad else 0)
