"""The fused pyramid kernel's own output, plane for plane and bit for bit.

The hot path never runs the two-pass stage kernels behind blurOctave, so their
parity says nothing about it.  Here the S+3 blur planes the fused kernel leaves in the
scratch arena are read back (sift_debug_read_planes) for every octave and compared with
the oracle's row pass, column pass and down-sampling (xblurMultiKernel, yblurKernel and
resizeKernel, Sift/src/excludes/cudaSiftD.cu:430-534, 544-559): float source, 8-bit
source, the stride-2 fetch of later octaves, both detector pre-filters, 2 and 3 scales,
ragged sizes, and BASELINE's batch of 256 images at 1080p with its strip geometry.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

BATCH_LANE0 = 8  # slot index of the batch entries' first launch lane (SIFT_MAX_SLOTS)


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _oracle_octaves(oracle, pr, img, w, n_oct):
    """Blur planes of every octave as the reference builds them: (planes, width) per octave."""
    out = []
    src = np.ascontiguousarray(img, np.float32)
    for _ in range(n_oct):
        blur = oracle.yblur(pr, oracle.xblur(pr, src, w), w)
        out.append((blur, w))
        src = oracle.downsample(blur[pr.n_scales], w)  # cudaSiftH.cu:50-52: even pixels of plane S
        src = np.nan_to_num(src, nan=0.0)  # pad columns: never read inside the width
        w = (w + 1) // 2
    return out


def _assert_octaves_equal(sift, oracle, pr, img, w, slot, image, n_oct, what):
    ref = _oracle_octaves(oracle, pr, img, w, n_oct)
    for o, (rb, ow) in enumerate(ref):
        got, gw = sift.debug_read_planes(slot, image, o)
        assert gw == ow and got.shape[:2] == rb.shape[:2], f"{what} octave {o}: geometry {got.shape} width {gw}"
        bad = _bits(got[..., :ow]) != _bits(rb[..., :ow])
        assert not bad.any(), (f"{what} octave {o}: {bad.sum()} of {bad.size} values differ, "
                               f"first at (plane,y,x) {np.argwhere(bad)[0]}")


@pytest.mark.parametrize(
    "w,h,S,O,nimg,profile,u8",
    [(640, 480, 3, 4, 1, 0, False),     # BASELINE configs[1]
     (800, 640, 2, 4, 2, 0, False),     # the reference's own image size and defaults
     (333, 129, 2, 3, 3, 0, False),     # ragged: width and height no multiple of anything
     (1000, 37, 2, 2, 1, 0, False),     # wider than tall, one strip
     (97, 70, 3, 2, 5, 0, True),        # 8-bit source, tiny later octaves
     (1024, 575, 2, 4, 2, 0, True),     # 8-bit source at the size of Sift/img/img3.ppm
     (640, 480, 3, 4, 2, 1, False),     # the plain pre-filter (profile lowe), 3 scales
     (801, 603, 2, 5, 4, 1, True),      # plain pre-filter, 8-bit, odd sizes down to 51x38
     (1920, 1080, 2, 5, 2, 0, False)],  # full size with short strips (few images per launch)
)
def test_fused_kernel_planes_match_the_oracle(sift, oracle_mod, w, h, S, O, nimg, profile, u8):
    kw = dict(n_scales=S, n_octaves=O, max_points=4000, profile=profile)
    sift.set_params(**kw)
    pr = oracle_mod.params(**kw)
    try:
        imgs = np.stack([sift.synth_image(w, h, seed=7000 + 13 * i + w, dense=bool(i & 1)) for i in range(nimg)])
        d = torch.from_numpy(imgs.astype(np.uint8) if u8 else imgs).cuda()  # the images hold integers
        sift.extract_batch(d, w)
        for i in sorted({0, nimg - 1}):
            slot, k = sift.debug_batch_slot(nimg, i)
            _assert_octaves_equal(sift, oracle_mod, pr, imgs[i], w, slot, k, O, f"{w}x{h} S={S} image {i}")
    finally:
        sift.set_params()


@pytest.mark.parametrize(
    "w,h,O,nimg,u8,min_w",
    [(800, 640, 4, 2, False, 1),      # every octave through the two-column kernel, down to 100x80
     (333, 129, 3, 3, False, 1),      # ragged: a last tile of 77 columns, widths below one tile
     (1024, 575, 4, 2, True, 1),      # 8-bit source
     (1920, 1080, 5, 3, False, 512),  # the bench's mix: wide octaves two columns per lane, narrow ones one
     (130, 40, 2, 1, False, 1)],
)
def test_two_columns_per_lane_planes_and_points_match_the_oracle(sift, oracle_mod, w, h, O, nimg, u8, min_w):
    """pyramid2_kernel (128-column tiles, two columns per lane): the same planes bit for bit, and the
    same keypoints through its pre-filter's queues."""
    kw = dict(n_scales=2, n_octaves=O, max_points=20000)
    sift.set_params(**kw)
    pr = oracle_mod.params(**kw)
    sift.lib().sift_set_pyramid_columns(2, min_w)
    try:
        imgs = np.stack([sift.synth_image(w, h, seed=8100 + 17 * i + w, dense=bool(i & 1)) for i in range(nimg)])
        d = torch.from_numpy(imgs.astype(np.uint8) if u8 else imgs).cuda()
        d_pts, counts = sift.extract_batch(d, w)
        for i in range(nimg):
            slot, k = sift.debug_batch_slot(nimg, i)
            _assert_octaves_equal(sift, oracle_mod, pr, imgs[i], w, slot, k, O, f"two columns {w}x{h} image {i}")
            ref, _, total = oracle_mod.extract(pr, imgs[i], w)
            assert total < 20000 and counts[i] == len(ref), f"image {i}: {counts[i]} points, oracle {len(ref)}"
            got = sift.points_to_numpy(d_pts, int(counts[i]), offset=i * 20000)
            assert oracle_mod.sort_points(got).tobytes() == oracle_mod.sort_points(ref).tobytes(), f"image {i}"
    finally:
        sift.lib().sift_set_pyramid_columns(1, 512)
        sift.set_params()


@pytest.mark.parametrize(
    "w,h,S,O,u8,pitch_extra",
    [(1920, 1080, 2, 5, False, 0),   # one 1080p image: every octave a single round of waves
     (640, 480, 3, 4, False, 0),     # three scales: six planes, six waves per tile
     (801, 603, 2, 5, True, 0),      # 8-bit source, odd sizes down to 51x38
     (333, 129, 3, 3, False, 5),     # ragged, pitched with garbage in the pad columns
     (97, 70, 2, 2, True, 0),
     (1000, 37, 2, 2, False, 0)],    # wider than tall: one strip of 37 rows
)
def test_plane_per_wave_kernel_planes_and_points_match_the_oracle(sift, oracle_mod, w, h, S, O, u8, pitch_extra):
    """pyramid_plane_kernel (one blur plane per wave; the pyramid pass of launch sets small enough for
    the dense route): the planes of every octave bit for bit against the oracle's row pass, column
    pass and down-sampling, and the keypoints; then the same call with the kernel switched off."""
    kw = dict(n_scales=S, n_octaves=O, max_points=30000)
    sift.set_params(**kw)
    pr = oracle_mod.params(**kw)
    L = sift.lib()
    try:
        L.sift_set_dense_route_pixels(4000000)
        img = sift.synth_image(w, h, seed=9100 + w, dense=True)
        ref, _, total = oracle_mod.extract(pr, img, w)
        assert 0 < total == len(ref)
        want = oracle_mod.sort_points(ref).tobytes()
        if u8:  # the 8-bit entry is a batch entry: a batch of one
            d = torch.from_numpy(img.astype(np.uint8)[None]).cuda()
            for split in (1, 0):
                L.sift_set_pyramid_split(split)
                d_pts, counts = sift.extract_batch(d, w)
                slot, k = sift.debug_batch_slot(1, 0)
                _assert_octaves_equal(sift, oracle_mod, pr, img, w, slot, k, O, f"split {split} u8 {w}x{h}")
                got = sift.points_to_numpy(d_pts, int(counts[0]))
                assert oracle_mod.sort_points(got).tobytes() == want, f"split {split}"
        else:
            pitch = w + pitch_extra
            host = oracle_mod.pitched(img[:, :w], pitch, fill=-1e30) if pitch_extra else img
            d = torch.from_numpy(host).cuda()
            for split in (1, 0):
                L.sift_set_pyramid_split(split)
                pts = sift.new_point_buffer()
                sift.initDeviceVariables()
                sift.extractSift(pts, d, w, pitch, h, None, 1, 0)
                n = sift.getPointCount(1)
                _assert_octaves_equal(sift, oracle_mod, pr, host, w, 1, 0, O, f"split {split} {w}x{h}")
                assert sum(sift.group_counts(1)[1]) == 0  # no queue was fed: the dense route
                assert oracle_mod.sort_points(sift.points_to_numpy(pts, n)).tobytes() == want, f"split {split}"
    finally:
        L.sift_set_pyramid_split(1)
        L.sift_set_dense_route_pixels(0)
        sift.set_params()


def test_fused_kernel_planes_through_extractSift_with_a_pitch(sift, oracle_mod):
    """The reference entry point on a pitched image whose pad columns hold garbage."""
    kw = dict(n_scales=2, n_octaves=4, max_points=20000)
    sift.set_params(**kw)
    pr = oracle_mod.params(**kw)
    w, h, pitch = 301, 203, 303
    host = oracle_mod.pitched(sift.synth_image(w, h, seed=9, dense=True)[:, :w], pitch, fill=-1e30)
    pts = sift.new_point_buffer()
    sift.initDeviceVariables()
    sift.extractSift(pts, torch.from_numpy(host).cuda(), w, pitch, h, None, 2, 0)
    assert sift.getPointCount(2) > 0
    _assert_octaves_equal(sift, oracle_mod, pr, host, w, 2, 0, 4, "extractSift 301x203 pitch 303")
    sift.set_params()


def test_fused_kernel_planes_with_the_doubled_first_octave(sift, oracle_mod):
    kw = dict(n_scales=2, n_octaves=3, max_points=30000, profile=1, upsample=1)
    sift.set_params(**kw)
    pr = oracle_mod.params(**kw)
    try:
        w, h = 200, 150
        img = sift.synth_image(w, h, seed=77, dense=True)
        sift.extract_batch(torch.from_numpy(img[None]).cuda(), w)
        up = oracle_mod.upsample2x(img, w)
        _assert_octaves_equal(sift, oracle_mod, pr, up, 2 * w, BATCH_LANE0, 0, 4, "up-sampled 400x300")
    finally:
        sift.set_params()


# ---- BASELINE configs[2]: 256 synthetic 1920x1080 images, seeds 1000..1255, 2 scales, 5 octaves ----
@pytest.fixture(scope="module")
def config3(sift, oracle_mod):
    from concurrent.futures import ThreadPoolExecutor

    w, h, n = 1920, 1080, 256
    kw = dict(n_scales=2, n_octaves=5, max_points=2500)
    sift.set_params(**kw)
    sift.lib().sift_set_batch_chunk(256)
    imgs = np.empty((n, h, w), np.float32)

    def one(i):
        sift.lib().sift_synth_image(imgs[i].ctypes.data, w, w, h, 1000 + i, 0)

    with ThreadPoolExecutor(16) as ex:
        list(ex.map(one, range(n)))
    d = torch.from_numpy(imgs).cuda()
    d_pts, counts = sift.extract_batch(d, w)
    yield dict(sift=sift, pr=oracle_mod.params(**kw), imgs=imgs, d_pts=d_pts, counts=counts, w=w, n=n)
    del d, d_pts
    sift.set_params()
    sift.shutdown()  # the 14 GB arena goes back


def test_config3_planes_at_the_bench_geometry(config3, oracle_mod):
    """The launch sets, strip heights and grids of the benchmark's batch."""
    c = config3
    for i in (0, 127, 128, 255):
        slot, k = c["sift"].debug_batch_slot(c["n"], i)
        _assert_octaves_equal(c["sift"], oracle_mod, c["pr"], c["imgs"][i], c["w"], slot, k, 5,
                              f"config 3 image {i}")


def test_config3_counts_of_all_256_and_points_of_16(config3, oracle_mod):
    c = config3
    sift, pr = c["sift"], c["pr"]
    _, ref_counts = oracle_mod.extract_batch(pr, c["imgs"], c["w"], n_threads=16)
    assert c["counts"].tolist() == ref_counts.tolist()
    assert ref_counts.max() < pr.max_points and ref_counts.sum() > 20000
    for i in (0, 1, 17, 34, 51, 68, 85, 102, 127, 128, 153, 170, 204, 221, 238, 255):
        ref, _, total = oracle_mod.extract(pr, c["imgs"][i], c["w"])
        assert total == len(ref) == ref_counts[i]
        got = sift.points_to_numpy(c["d_pts"], int(c["counts"][i]), offset=i * pr.max_points)
        a = oracle_mod.sort_points(got.view(oracle_mod.POINT_DTYPE))
        b = oracle_mod.sort_points(ref)
        assert a.tobytes() == b.tobytes(), f"image {i}: points differ from the oracle"
