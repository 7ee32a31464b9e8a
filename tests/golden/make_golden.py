#!/usr/bin/env python3
"""Regenerates the fixtures in this directory.  Run from the repository root in
the build container (needs /root/reference for the image-derived fixtures):

    python tests/golden/make_golden.py

What each file pins:
  tables_S2.txt / tables_S3.txt  hex-float Gaussian tables from the CPU oracle, so a libm
                                 change on another machine cannot silently move them.
  gray_img3_crop.npz             a 96x64 crop of the reference's Sift/img/img3.ppm (RGB) and the
                                 same crop of Sift/result/result.ppm, the reference's only
                                 known-answer file: result == OpenCV grayscale load of img3.
  real_img1_crop.npz             a 320x256 grey crop of Sift/img/img1.ppm (data) and the
                                 oracle's keypoints for it (S=2, O=3): real-image parity case
                                 for machines without /root/reference.  Oracle-generated.
  extract_640x480_S3.npz         oracle keypoints for the synthetic 640x480, 3 scales,
                                 4 octaves, seed 1 image.  Oracle-generated: a regression
                                 anchor, not a reference output (the reference cannot run here).
  extract_lowe_400x300_up.npz    the same for profile "lowe" with the doubled first octave
                                 (400x300, 2 scales, 3 + 1 octaves, seed 3): an anchor against the
                                 build's own definition drifting; nothing in the reference defines it.

  config1_images.npz             the reference's three sample images at full size as 8-bit grey
                                 (Sift/img/img1.ppm, img2.ppm: 800x640; img3.ppm: 1024x575; grey =
                                 image::imload's weights, which Sift/result/result.ppm pins for
                                 img3): inputs of the configuration the reference itself runs
                                 (2 scales, 4 octaves, cap 2500).  Data only; expected points come
                                 from the oracle at test time.

`python tests/golden/make_golden.py lowe` rewrites only the lowe fixture, `... config1` only the images.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
import sift_amd  # noqa: E402
from pnm import read_pnm  # noqa: E402

REF = "/root/reference/Sift"


def hexrow(a):
    return " ".join(float(v).hex() for v in np.asarray(a, np.float32).ravel())


def lowe_fixture():
    kw = dict(n_scales=2, n_octaves=3, max_points=30000, profile=1, upsample=1)
    pr = oracle.params(**kw)
    img = sift_amd.synth_image(400, 300, seed=3, dense=True)
    pts, meta, total = oracle.extract(pr, img, 400)
    assert total == len(pts)
    np.savez_compressed(os.path.join(HERE, "extract_lowe_400x300_up.npz"),
                        image_sha256=hashlib.sha256(img.tobytes()).hexdigest(),
                        points=oracle.sort_points(pts), quantized=oracle.quantize_descriptors(oracle.sort_points(pts)),
                        n_scales=2, n_octaves=3, seed=3, dense=1)
    print("lowe 400x300 up-sampled keypoints:", len(pts))


def config1_fixture():
    out = {}
    for k, name in ((1, "img1"), (2, "img2"), (3, "img3")):
        g = oracle.gray_q14(read_pnm(os.path.join(REF, "img", name + ".ppm")))
        out[f"gray{k}"] = g.astype(np.uint8)
    res = read_pnm(os.path.join(REF, "result/result.ppm"))[..., 0]
    assert np.array_equal(out["gray3"], res)  # the reference's own grey version of img3
    np.savez_compressed(os.path.join(HERE, "config1_images.npz"), **out)
    print("config 1 images:", {k: v.shape for k, v in out.items()})


def main():
    if sys.argv[1:] == ["lowe"]:
        lowe_fixture()
        return
    if sys.argv[1:] == ["config1"]:
        config1_fixture()
        return
    config1_fixture()
    lowe_fixture()
    for S in (2, 3):
        pr = oracle.params(n_scales=S)
        blur, wnd, desc = oracle.tables(pr)
        c32, s32 = oracle.rotation_table()
        with open(os.path.join(HERE, f"tables_S{S}.txt"), "w") as f:
            f.write("blur " + hexrow(blur) + "\n")
            f.write("wnd " + hexrow(wnd) + "\n")
            f.write("desc " + hexrow(desc) + "\n")
            f.write("cos32 " + hexrow(c32) + "\n")
            f.write("sin32 " + hexrow(s32) + "\n")
            f.write("scale " + hexrow([oracle.scale_value(pr, i, o) for o in range(6) for i in range(S)]) + "\n")

    rgb = read_pnm(os.path.join(REF, "img/img3.ppm"))
    res = read_pnm(os.path.join(REF, "result/result.ppm"))[..., 0]
    y0, x0 = 200, 400
    np.savez_compressed(
        os.path.join(HERE, "gray_img3_crop.npz"),
        rgb=rgb[y0:y0 + 64, x0:x0 + 96], gray=res[y0:y0 + 64, x0:x0 + 96],
        full_gray_sha256=hashlib.sha256(res.tobytes()).hexdigest(),
        full_rgb_sha256=hashlib.sha256(rgb.tobytes()).hexdigest())

    g1 = oracle.gray_q14(read_pnm(os.path.join(REF, "img/img1.ppm")))
    crop = np.ascontiguousarray(g1[100:356, 200:520])
    pr = oracle.params(n_scales=2, n_octaves=3, max_points=20000)
    pts, meta, total = oracle.extract(pr, crop.astype(np.float32), crop.shape[1])
    assert total == len(pts)
    np.savez_compressed(os.path.join(HERE, "real_img1_crop.npz"), gray=crop,
                        points=oracle.sort_points(pts), n_scales=2, n_octaves=3)
    print("real crop keypoints:", len(pts))

    pr = oracle.params(n_scales=3, n_octaves=4, max_points=20000)
    img = sift_amd.synth_image(640, 480, seed=1, dense=True)
    pts, meta, total = oracle.extract(pr, img, 640)
    assert total == len(pts)
    np.savez_compressed(os.path.join(HERE, "extract_640x480_S3.npz"),
                        image_sha256=hashlib.sha256(img.tobytes()).hexdigest(),
                        points=oracle.sort_points(pts), n_scales=3, n_octaves=4, seed=1, dense=1)
    print("synthetic 640x480 keypoints:", len(pts))


if __name__ == "__main__":
    main()
