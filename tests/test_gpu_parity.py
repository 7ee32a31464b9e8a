"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs.  Bit-exact everywhere (float sums have one fixed order
on both sides); keypoint lists are compared as sets (append order is atomic)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _params(sift, oracle, **kw):
    sift.set_params(**kw)
    return oracle.params(**kw)


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _assert_planes_equal(gpu, ref, w, what):
    g = gpu.cpu().numpy()[..., :w]
    r = ref[..., :w]
    bad = _bits(g) != _bits(r)
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} values differ, first at {np.argwhere(bad)[0]}"


@pytest.mark.parametrize("w,h,S", [(640, 480, 3), (800, 640, 2), (333, 129, 2), (64, 40, 3), (1000, 37, 2)])
def test_pyramid_stages_bit_exact(sift, oracle_mod, w, h, S):
    pr = _params(sift, oracle_mod, n_scales=S)
    pitch = (w + 127) // 128 * 128
    img = sift.synth_image(w, h, seed=1 + w, pitch=pitch)
    host = oracle_mod.pitched(img[:, :w], pitch)
    d_img = torch.from_numpy(host).cuda()
    blur = sift.blurOctave(d_img, w)
    ref_blur = oracle_mod.yblur(pr, oracle_mod.xblur(pr, host, w), w)
    _assert_planes_equal(blur, ref_blur, w, "blur")
    _assert_planes_equal(sift.computeDiffOfGauss(blur, w), oracle_mod.dog(pr, ref_blur, w), w, "DoG")
    _assert_planes_equal(sift.computeHessian(blur, w), oracle_mod.hessian(pr, ref_blur, w), w, "Hessian")
    _assert_planes_equal(sift.computeGradient(blur, w), oracle_mod.gradient(pr, ref_blur, w), w, "gradient")


def test_pyramid_stages_on_random_shapes(sift, oracle_mod):
    """The stage entries (blurOctave, computeDiffOfGauss, computeHessian, computeGradient) on
    random sizes and pitches (any multiple of 4 floats), bit for bit."""
    rng = np.random.default_rng(31)
    for case in range(12):
        w, h, S = int(rng.integers(8, 900)), int(rng.integers(8, 500)), int(rng.integers(2, 4))
        pitch = (w + 3) // 4 * 4 + 4 * int(rng.integers(0, 20))
        pr = _params(sift, oracle_mod, n_scales=S)
        img = sift.synth_image(w, h, seed=900 + case, pitch=pitch, dense=bool(case & 1))
        host = oracle_mod.pitched(img[:, :w], pitch)
        blur = sift.blurOctave(torch.from_numpy(host).cuda(), w)
        ref_blur = oracle_mod.yblur(pr, oracle_mod.xblur(pr, host, w), w)
        what = f"case {case}: {w}x{h} pitch {pitch} S={S} "
        _assert_planes_equal(blur, ref_blur, w, what + "blur")
        _assert_planes_equal(sift.computeDiffOfGauss(blur, w), oracle_mod.dog(pr, ref_blur, w), w, what + "DoG")
        _assert_planes_equal(sift.computeHessian(blur, w), oracle_mod.hessian(pr, ref_blur, w), w, what + "Hessian")
        _assert_planes_equal(sift.computeGradient(blur, w), oracle_mod.gradient(pr, ref_blur, w), w, what + "gradient")
    sift.set_params()


DENSE_ROUTE_DEFAULT = 2200000  # sift_set_dense_route_pixels' default (include/sift_amd.h)


def _extract_gpu(sift, img_host, w, both=False):
    """One image through the reference's call sequence, twice: by the queue route (pre-filter in the
    pyramid pass, candidate verification) and by the route the library takes by default for a job
    this small (dense detector alone).  The two must give the same points."""
    import oracle

    h, pitch = img_host.shape
    d_img = torch.from_numpy(np.nan_to_num(img_host, nan=-1e30)).cuda()
    outs = []
    try:
        for route in (DENSE_ROUTE_DEFAULT, 0):
            sift.lib().sift_set_dense_route_pixels(route)
            pts = sift.new_point_buffer()
            sift.initDeviceConstant()
            sift.initDeviceVariables()
            sift.extractSift(pts, d_img, w, pitch, h, None, 0, 0)
            n = sift.getPointCount(0)
            outs.append(sift.points_to_numpy(pts, n))
    finally:
        sift.lib().sift_set_dense_route_pixels(0)
    if sift.get_params().max_points > max(len(o) for o in outs):  # (over the cap the survivors are any subset)
        assert oracle.sort_points(outs[0]).tobytes() == oracle.sort_points(outs[1]).tobytes(), \
            "dense route and queue route disagree"
    return outs if both else outs[1]


def _assert_capped_like_the_reference(oracle_mod, got, every, cap):
    """Over the cap the reference keeps its groups in order: (octave, scale) groups are filled one
    after the other and everything after the cap is reached is skipped (cudaSiftH.cu:12-13, 82), so
    the output is every point of the groups before the one that overflows plus cap - (their number)
    points of that group, any of them (the append order inside a group is an atomic's).  `every` is
    the oracle's list without a cap; a point's group is known from its scale value, which is unique
    per (octave, scale) (cudaSiftH.cu:94-95)."""
    got = got.view(oracle_mod.POINT_DTYPE)
    assert len(got) == min(cap, len(every))
    scales = sorted(set(every["scale"].tolist()))  # ascending scale = group order
    have = {r.tobytes() for r in every}
    assert all(r.tobytes() in have for r in got), "a point that is not in the uncapped list"
    assert len({r.tobytes() for r in got}) == len(got), "a point twice"
    room = cap
    for sc in scales:
        n_ref = int((every["scale"] == sc).sum())
        n_got = int((got["scale"] == sc).sum())
        assert n_got == min(n_ref, room), f"group of scale {sc}: {n_got} points, expected {min(n_ref, room)}"
        room -= n_got


def _assert_points_equal(oracle_mod, got, ref):
    assert len(got) == len(ref), f"count {len(got)} vs oracle {len(ref)}"
    a = oracle_mod.sort_points(got)
    b = oracle_mod.sort_points(ref)
    for f in ("xpos", "ypos", "scale", "orientation"):
        assert np.array_equal(_bits(a[f]), _bits(b[f])), f
    bad = (_bits(a["data"]) != _bits(b["data"])).any(axis=1)
    assert not bad.any(), f"{bad.sum()} descriptors differ"


@pytest.mark.parametrize(
    "w,h,S,O,seed,dense",
    [(640, 480, 3, 4, 1, False), (800, 640, 2, 4, 2, False), (1024, 575, 2, 4, 3, True),
     (200, 150, 2, 3, 4, True), (97, 70, 3, 2, 5, True), (640, 480, 3, 4, 6, True)],
)
def test_extract_matches_oracle(sift, oracle_mod, w, h, S, O, seed, dense):
    pr = _params(sift, oracle_mod, n_scales=S, n_octaves=O, max_points=20000)
    pitch = (w + 127) // 128 * 128
    host = oracle_mod.pitched(sift.synth_image(w, h, seed=seed, dense=dense)[:, :w], pitch)
    ref, meta, total = oracle_mod.extract(pr, host, w)
    assert total == len(ref) < pr.max_points
    assert len(ref) > 0
    got = _extract_gpu(sift, host, w)
    _assert_points_equal(oracle_mod, got, ref)


import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_real_image_crop_against_committed_golden(sift, oracle_mod):
    """A crop of the reference's own Sift/img/img1.ppm (committed as data)."""
    z = np.load(os.path.join(GOLD, "real_img1_crop.npz"))
    _params(sift, oracle_mod, n_scales=int(z["n_scales"]), n_octaves=int(z["n_octaves"]), max_points=20000)
    g = z["gray"].astype(np.float32)
    got = _extract_gpu(sift, g, g.shape[1])
    assert oracle_mod.sort_points(got.view(oracle_mod.POINT_DTYPE)).tobytes() == z["points"].tobytes()


@pytest.mark.parametrize("key", ["gray1", "gray2", "gray3"])
def test_reference_sample_images_at_full_size(sift, oracle_mod, key):
    """The configuration the reference itself runs (Sift/src/utils.h:4-19: 2 scales, 4 octaves, cap
    2500) on its own three sample images at full size (Sift/img/img1.ppm, img2.ppm, img3.ppm,
    committed as 8-bit grey data).  With a cap that holds everything the points equal the oracle's
    as sets, bit for bit, through the single-image, the batch and the 8-bit batch entries; at the
    reference's cap, which the two 800x640 photographs exceed, the output is a cap-sized subset of
    that list."""
    g8 = np.load(os.path.join(GOLD, "config1_images.npz"))[key]
    g = g8.astype(np.float32)
    h, w = g.shape
    big = 60000
    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=4, max_points=big)
    ref, _, total = oracle_mod.extract(pr, g, w)
    assert 600 < total == len(ref) < big
    want = oracle_mod.sort_points(ref).tobytes()
    got = _extract_gpu(sift, g, w)
    assert oracle_mod.sort_points(got.view(oracle_mod.POINT_DTYPE)).tobytes() == want
    for stack in (g[None], g8[None]):
        pts, counts = sift.extract_batch(torch.from_numpy(np.ascontiguousarray(stack)).cuda(), w)
        got = sift.points_to_numpy(pts, int(counts[0])).view(oracle_mod.POINT_DTYPE)
        assert oracle_mod.sort_points(got).tobytes() == want
    _params(sift, oracle_mod, n_scales=2, n_octaves=4, max_points=2500)
    for got in _extract_gpu(sift, g, w, both=True):  # the dense route and the queue route
        _assert_capped_like_the_reference(oracle_mod, got, ref, 2500)
    for stack in (g[None], g8[None]):
        pts, counts = sift.extract_batch(torch.from_numpy(np.ascontiguousarray(stack)).cuda(), w)
        _assert_capped_like_the_reference(oracle_mod, sift.points_to_numpy(pts, int(counts[0])), ref, 2500)
    sift.set_params()


def test_lowe_against_committed_golden(sift, oracle_mod):
    """Profile "lowe" with the doubled first octave against the committed fixture: refined
    points and their 8-bit descriptors, byte for byte."""
    import hashlib

    z = np.load(os.path.join(GOLD, "extract_lowe_400x300_up.npz"))
    sift.set_params(n_scales=int(z["n_scales"]), n_octaves=int(z["n_octaves"]), max_points=30000, profile=1, upsample=1)
    img = sift.synth_image(400, 300, seed=int(z["seed"]), dense=bool(z["dense"]))
    if hashlib.sha256(img.tobytes()).hexdigest() != str(z["image_sha256"]):
        pytest.skip("libm differs from the machine that made the fixture: the synthetic image moved")
    try:
        pts, counts = sift.extract_batch(torch.from_numpy(img[None]).cuda(), 400)
        n = int(counts[0])
        q = sift.quantize_descriptors(pts, n).cpu().numpy()
    finally:
        sift.set_params()
    got = sift.points_to_numpy(pts, n).view(oracle_mod.POINT_DTYPE)
    assert oracle_mod.sort_points(got).tobytes() == z["points"].tobytes()
    # descriptors quantised on the GPU, put into the fixture's order through the same sort
    srt = oracle_mod.sort_points(got)
    key = {r[["xpos", "ypos", "scale", "orientation"]].tobytes(): i for i, r in enumerate(srt)}
    perm = np.array([key[r[["xpos", "ypos", "scale", "orientation"]].tobytes()] for r in got])
    assert len(set(perm.tolist())) == n
    out = np.empty_like(q)
    out[perm] = q
    assert np.array_equal(out, z["quantized"])


def test_unaligned_pitch_and_odd_sizes(sift, oracle_mod):
    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=3, max_points=20000)
    w, h, pitch = 301, 203, 303  # pitch not a multiple of 4: the scalar staging path
    host = oracle_mod.pitched(sift.synth_image(w, h, seed=9, dense=True)[:, :w], pitch)
    ref, _, total = oracle_mod.extract(pr, host, w)
    assert total == len(ref) > 20
    _assert_points_equal(oracle_mod, _extract_gpu(sift, host, w), ref)


def test_counts_accumulate_until_reset_and_slots_are_independent(sift, oracle_mod):
    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=2, max_points=5000)
    a = sift.synth_image(256, 192, seed=21, dense=True)
    b = sift.synth_image(320, 200, seed=22, dense=True)
    ra, _, _ = oracle_mod.extract(pr, a, 256)
    rb, _, _ = oracle_mod.extract(pr, b, 320)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    p0, p1 = sift.new_point_buffer(), sift.new_point_buffer()
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    sift.initDeviceConstant()
    sift.initDeviceVariables()
    sift.extractSift(p0, da, 256, 256, 192, s0, 0, 0)
    sift.extractSift(p1, db, 320, 320, 200, s1, 1, 0)
    sift.extractSift(p0, db, 320, 320, 200, s0, 0, 0)  # appends behind the first image's points
    n0, n1 = sift.getPointCount(0), sift.getPointCount(1)
    assert n1 == len(rb) and n0 == len(ra) + len(rb)
    _assert_points_equal(oracle_mod, sift.points_to_numpy(p1, n1), rb)
    _assert_points_equal(oracle_mod, sift.points_to_numpy(p0, len(ra)), ra)
    _assert_points_equal(oracle_mod, sift.points_to_numpy(p0, len(rb), offset=len(ra)), rb)
    sift.initDeviceVariables()
    assert sift.getPointCount(0) == 0 and sift.getPointCount(1) == 0


@pytest.mark.parametrize("queue_cap,cap,S", [(65536, 256, 2), (0, 256, 2), (16, 256, 2), (0, 40, 2), (16, 300, 2),
                                             (0, 100, 3), (16, 350, 3)])
def test_cap_overflow_keeps_a_subset_and_clamps_the_count(sift, oracle_mod, queue_cap, cap, S):
    """More keypoints than max_points, through the queue path and through the dense detector
    (which stops working on an image once a group has reached the cap)."""
    w, h = 640, 480
    img = sift.synth_image(w, h, seed=6, dense=True)
    full, _, _ = oracle_mod.extract(oracle_mod.params(max_points=20000, n_scales=S), img, w)
    assert len(full) >= 300
    sift.set_params(max_points=cap, n_scales=S)
    sift.lib().sift_set_queue_cap(queue_cap)
    try:
        got = _extract_gpu(sift, img, w)
    finally:
        sift.lib().sift_set_queue_cap(65536)
    assert len(got) == min(cap, len(full))
    want = {p.tobytes() for p in full}
    assert all(p.tobytes() in want for p in got), "every kept point is a full-run point, bit for bit"
    # whole groups in front of the overflowing one are complete (the reference skips later scales)
    first = full["scale"] == full["scale"][0]
    if first.sum() < cap:
        assert (got["scale"] == full["scale"][0]).sum() == first.sum()


@pytest.mark.parametrize("queue_cap,cap", [(16, 60000), (65536, 60000), (0, 60000), (65536, 2500)])
def test_cap_reached_while_the_dense_detector_runs(sift, oracle_mod, queue_cap, cap):
    """A large image with a candidate nearly everywhere (contrast threshold 1, three scales):
    several groups of the first octave reach the cap one after the other while thousands of
    detector blocks are at work, and every block has to act on one consistent view of which
    groups are full.  The output is cap points, all of them the oracle's."""
    w, h = 2949, 1709
    kw = dict(n_scales=3, n_octaves=4, extrema_thresh=1.0)
    img = sift.synth_image(w, h, seed=196340271, dense=True)
    every, _, total = oracle_mod.extract(oracle_mod.params(max_points=400000, **kw), img, w)
    assert total == len(every) > 300000
    sift.set_params(max_points=cap, **kw)
    sift.lib().sift_set_queue_cap(queue_cap)
    try:
        for _ in range(3):  # the interleaving differs from run to run
            for got in _extract_gpu(sift, img, w, both=True):  # (one image: with and without the dense route)
                assert len(got) == cap
                _assert_capped_like_the_reference(oracle_mod, got, every, cap)
    finally:
        sift.lib().sift_set_queue_cap(65536)
        sift.set_params()
    sift.shutdown()


def test_start_octave_argument(sift, oracle_mod):
    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=4, max_points=20000)
    img = sift.synth_image(400, 300, seed=31, dense=True)
    ref, _, _ = oracle_mod.extract(pr, img, 400, octave0=2)
    assert len(ref) > 0
    d = torch.from_numpy(img).cuda()
    pts = sift.new_point_buffer()
    sift.initDeviceVariables()
    sift.extractSift(pts, d, 400, 400, 300, None, 3, 2)
    n = sift.getPointCount(3)
    _assert_points_equal(oracle_mod, sift.points_to_numpy(pts, n), ref)
    sift.extractSift(pts, d, 400, 400, 300, None, 3, 4)  # octaveIdx == n_octaves: nothing to do
    assert sift.getPointCount(3) == n


def test_batch_entry_matches_per_image_oracle(sift, oracle_mod):
    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=4, max_points=3000)
    n, w, h = 11, 480, 270
    imgs = np.stack([sift.synth_image(w, h, seed=100 + i, dense=(i % 2 == 0)) for i in range(n)])
    ref_pts, ref_counts = oracle_mod.extract_batch(pr, imgs, w, n_threads=8)
    sift.lib().sift_set_batch_chunk(4)  # 3 launch sets: 4 + 4 + 3
    d_pts, counts, meta = sift.extract_batch(torch.from_numpy(imgs).cuda(), w, want_meta=True)
    sift.lib().sift_set_batch_chunk(256)
    assert counts.tolist() == ref_counts.tolist() and counts.sum() > 100
    meta = meta.cpu().numpy().reshape(n, pr.max_points, 2)
    for i in range(n):
        got = sift.points_to_numpy(d_pts, int(counts[i]), offset=i * pr.max_points)
        _assert_points_equal(oracle_mod, got, ref_pts[i][: ref_counts[i]])
        m = meta[i, : counts[i]]
        sv = np.array([oracle_mod.scale_value(pr, int(s), int(o)) for o, s in m], np.float32)
        assert np.array_equal(sv, got["scale"]), "meta = (octave, scale index) of each point"
        assert np.all(np.diff(m[:, 0].astype(int) * 8 + m[:, 1]) >= 0), "grouped by octave then scale"


def test_full_size_1080p_batch_properties(sift, oracle_mod):
    """BASELINE size: 1920x1080, 2 scales, 5 octaves.  One image is checked bit for bit
    against the oracle; the batch is checked through size-independent properties."""
    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=5, max_points=2500)
    w, h, n = 1920, 1080, 6
    imgs = np.stack([sift.synth_image(w, h, seed=1000 + i, dense=(i == 5)) for i in range(n)])
    d = torch.from_numpy(imgs).cuda()
    d_pts, counts = sift.extract_batch(d, w)
    ref, _, total = oracle_mod.extract(pr, imgs[0], w)
    assert total == len(ref)
    _assert_points_equal(oracle_mod, sift.points_to_numpy(d_pts, int(counts[0])), ref)
    ref5, _, _ = oracle_mod.extract(pr, imgs[5], w)
    _assert_points_equal(oracle_mod, sift.points_to_numpy(d_pts, int(counts[5]), offset=5 * 2500), ref5)
    # idempotence: the same batch again gives the same bytes per image once sorted
    d_pts2, counts2 = sift.extract_batch(d, w)
    assert counts.tolist() == counts2.tolist()
    for i in range(n):
        a = sift.points_to_numpy(d_pts, int(counts[i]), offset=i * 2500)
        b = sift.points_to_numpy(d_pts2, int(counts[i]), offset=i * 2500)
        assert oracle_mod.sort_points(a).tobytes() == oracle_mod.sort_points(b).tobytes()
        assert np.allclose(a["data"].sum(1), 1, atol=1e-5) and np.all(a["orientation"] % 11.25 == 0)
    # permutation: image order in the batch does not matter
    d_pts3, counts3 = sift.extract_batch(torch.from_numpy(imgs[::-1].copy()).cuda(), w)
    assert counts3.tolist() == counts.tolist()[::-1]


def test_bad_arguments_are_reported_not_fatal(sift):
    sift.set_params()
    pts = sift.new_point_buffer()
    d = torch.zeros((10, 16), dtype=torch.float32, device="cuda")
    with pytest.raises(sift.SiftError):
        sift.extractSift(pts, d, 32, 16, 10, None, 0, 0)  # pitch < width
    with pytest.raises(sift.SiftError):
        sift.extractSift(pts, d, 16, 16, 10, None, 99, 0)  # slot out of range
    sift.initDeviceVariables()
    sift.extractSift(pts, d, 16, 16, 10, None, 0, 0)  # too small for any keypoint: fine, zero points
    assert sift.getPointCount(0) == 0


@pytest.mark.parametrize("queue_cap", [0, 16, 300])
@pytest.mark.parametrize("cap", [20000, 4000])
def test_detection_paths_agree(sift, oracle_mod, queue_cap, cap):
    """Queue capacity 0 = dense detector only; tiny capacities force the overflow fall-back for
    some (octave, scale) groups and not others.  The fall-back walks the image block by block
    (large cap) or group by group (cap 4000).  All paths give the oracle's points."""
    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=4, max_points=cap)
    w, h = 640, 480
    img = sift.synth_image(w, h, seed=6, dense=True)
    ref, meta, _ = oracle_mod.extract(pr, img, w)
    assert (meta[:, 0] == 0).sum() > 100 > (meta[:, 0] == 1).sum() > 0
    assert (ref["ypos"].astype(int) % 32 >= 30).sum() > 0, "covers the aliased-centre rows"
    sift.lib().sift_set_queue_cap(queue_cap)
    try:
        got = _extract_gpu(sift, img, w)
    finally:
        sift.lib().sift_set_queue_cap(65536)
    _assert_points_equal(oracle_mod, got, ref)


@pytest.mark.parametrize("thresh", [3.0, 0.5])
def test_many_candidates_per_strip(sift, oracle_mod, thresh):
    """A low contrast threshold on a tall image: every 64-column strip of the pyramid kernel
    collects far more candidates than its in-LDS list holds between flushes (threshold 3) or
    than one chunk of rows may add (threshold 0.5: the list overflows and the dense detector
    takes the group over).  The result is still the oracle's."""
    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=3, max_points=60000, extrema_thresh=thresh)
    w, h = 192, 1500
    img = sift.synth_image(w, h, seed=77, dense=True)
    ref, _, total = oracle_mod.extract(pr, img, w)
    assert total == len(ref) > 1000
    got = _extract_gpu(sift, img, w)
    _assert_points_equal(oracle_mod, got, ref)


def test_sparse_and_dense_detectors_agree_on_random_shapes(sift, oracle_mod):
    """The queue-driven detector (contrast pre-filter in the pyramid kernel, aliased-row
    pre-filter, verify) against the dense detector over random sizes, scales and contrast
    thresholds.  The dense detector is the one the oracle comparisons above pin."""
    rng = np.random.default_rng(5)
    for case in range(16):
        w, h = int(rng.integers(40, 700)), int(rng.integers(40, 700))
        S = int(rng.integers(2, 4))
        thr = float(rng.choice([1.0, 3.0, 10.0]))
        sift.set_params(n_scales=S, n_octaves=4, max_points=60000, extrema_thresh=thr)
        img = sift.synth_image(w, h, seed=300 + case, dense=True)
        sift.lib().sift_set_queue_cap(0)
        try:
            dense = _extract_gpu(sift, img, w)
        finally:
            sift.lib().sift_set_queue_cap(65536)
        sparse = _extract_gpu(sift, img, w)
        assert len(dense) == len(sparse), (case, w, h, S, thr, len(dense), len(sparse))
        assert oracle_mod.sort_points(dense).tobytes() == oracle_mod.sort_points(sparse).tobytes(), (case, w, h, S, thr)
    assert len(dense) > 0


def test_lowe_queue_and_dense_paths_agree_on_random_shapes(sift, oracle_mod):
    """Profile "lowe": the queue path (the pyramid kernel's in-plane extremum pre-filter across
    lanes, rows and strip borders, then the refinement kernel) against the dense detector and the
    oracle, over random sizes, scales, thresholds and the doubled first octave."""
    rng = np.random.default_rng(11)
    total = 0
    for case in range(14):
        w, h = int(rng.integers(40, 700)), int(rng.integers(40, 900))
        S = int(rng.integers(2, 4))
        thr = float(rng.choice([2.0, 10.0, 25.0]))
        up = int(case % 4 == 3)
        kw = dict(n_scales=S, n_octaves=3, max_points=60000, extrema_thresh=thr, profile=1, upsample=up)
        pr = _params(sift, oracle_mod, **kw)
        img = sift.synth_image(w, h, seed=700 + case, dense=True)
        ref, _, found = oracle_mod.extract(pr, img, w)
        assert found == len(ref)
        try:
            sift.lib().sift_set_queue_cap(0)
            dense = _extract_gpu(sift, img, w)
            sift.lib().sift_set_queue_cap(65536)
            sparse = _extract_gpu(sift, img, w)
        finally:
            sift.lib().sift_set_queue_cap(65536)
            sift.set_params()
        _assert_points_equal(oracle_mod, dense, ref)
        _assert_points_equal(oracle_mod, sparse, ref)
        total += len(ref)
    assert total > 2000


def test_u8_batch_entry_equals_float_entry(sift, oracle_mod):
    """8-bit images converted by the pyramid kernel's fetch: same bytes out as the float entry on
    the converted images, which the oracle pins; odd width and a pitch that is not a multiple of 4."""
    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=4, max_points=4000)
    n, w, h, pitch = 5, 333, 250, 335
    imgs = np.stack([sift.synth_image(w, h, seed=500 + i, dense=True) for i in range(n)])
    assert np.array_equal(imgs, np.round(imgs)) and imgs.min() >= 0 and imgs.max() <= 255
    u8 = np.zeros((n, h, pitch), np.uint8)
    u8[:, :, :w] = imgs.astype(np.uint8)
    pts8, counts8 = sift.extract_batch(torch.from_numpy(u8).cuda(), w)
    ptsf, countsf = sift.extract_batch(torch.from_numpy(imgs).cuda(), w)
    assert counts8.tolist() == countsf.tolist() and counts8.sum() > 200
    ref, _, _ = oracle_mod.extract(pr, imgs[0], w)
    _assert_points_equal(oracle_mod, sift.points_to_numpy(pts8, int(counts8[0])), ref)
    for i in range(n):
        a = sift.points_to_numpy(pts8, int(counts8[i]), offset=i * pr.max_points)
        b = sift.points_to_numpy(ptsf, int(countsf[i]), offset=i * pr.max_points)
        assert oracle_mod.sort_points(a).tobytes() == oracle_mod.sort_points(b).tobytes()


def test_host_pipeline_writes_the_oracles_text_files(sift, oracle_mod, tmp_path):
    """sift() (the counterpart of Sift/src/sift.cu:64-143): PGM/PPM files in, one .sift.txt per
    image out, images taken two at a time on two streams.  Lines are compared as sets (the order
    inside an (octave, scale) group is an atomic's) against the oracle's formatted points."""
    import ctypes as C
    from pnm import write_pnm

    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=4, max_points=2500)
    src = tmp_path / "src"
    dst = tmp_path / "dst"
    src.mkdir()
    dst.mkdir()
    paths, grays = [], []
    # seven files: sizes that grow and shrink from pair to pair (the pipeline keeps its buffers),
    # an image too small for any keypoint, and an odd one left over at the end
    for i, (w, h) in enumerate([(320, 200), (257, 131), (400, 300), (64, 48), (150, 333), (30, 20), (401, 299)]):
        img = sift.synth_image(w, h, seed=700 + i, dense=True).astype(np.uint8)
        if i == 2:  # a colour file: the loader's Q14 grey conversion is part of the pipeline
            rgb = np.stack([img, np.roll(img, 3, axis=1), np.roll(img, 5, axis=0)], axis=-1)
            p = src / f"img{i}.ppm"
            write_pnm(str(p), rgb)
            grays.append(oracle_mod.gray_q14(rgb).astype(np.float32))
        else:
            p = src / f"img{i}.pgm"
            write_pnm(str(p), img)
            grays.append(img.astype(np.float32))
        paths.append(str(p))
    arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
    for drop in (0, 1):
        sift.lib().sift_pipeline_drop_last_point(drop)
        try:
            assert sift.lib().sift(os.fsencode(str(dst)), arr, len(paths)) == 0, sift.last_error()
        finally:
            sift.lib().sift_pipeline_drop_last_point(0)
        for p, g in zip(paths, grays):
            ref, _, total = oracle_mod.extract(pr, g, g.shape[1])
            assert total == len(ref) and (len(ref) > 10 or g.size < 10000)
            want = oracle_mod.format_points(ref).splitlines()
            got = open(dst / (os.path.basename(p) + ".sift.txt")).read().splitlines()
            if drop:  # the reference hands its writer count - 1 (sift.cu:125)
                assert len(got) == max(len(want) - 1, 0) and set(got) <= set(want)
            else:  # the writer's order is the oracle's canonical one: scale, row, column
                assert got == oracle_mod.format_points(oracle_mod.sort_points(ref)).splitlines()


def test_large_8k_image_against_oracle(sift, oracle_mod):
    """One dense 7680x4320 image, 6 octaves, cap 2^20: 31 k keypoints.  The first octave's
    candidate queues overflow at this size, so the dense detector takes those groups over while
    the later octaves stay on the queue path."""
    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=6, max_points=1 << 20)
    w, h = 7680, 4320
    img = sift.synth_image(w, h, seed=9000, dense=True)
    ref, meta, total = oracle_mod.extract(pr, img, w)
    assert total == len(ref) > 30000 and (meta[:, 0] == 1).sum() > 1000
    got = _extract_gpu(sift, img, w)
    _assert_points_equal(oracle_mod, got, ref)
    sift.shutdown()  # give the 1.2 GB of arena and lists back before the next test


@pytest.mark.parametrize("profile", [1, 0])
def test_upsampled_8k_image_against_oracle(sift, oracle_mod, profile):
    """The largest configuration, 15360x8640 with 177 Mpx of pyramid and cap 2^20: a dense
    7680x4320 image doubled by the library (profile "lowe", 6 + 1 octaves), and the doubled image
    handed in as it is (profile "ref", 6 octaves)."""
    w, h = 7680, 4320
    img = sift.synth_image(w, h, seed=9001, dense=True)
    if profile == 1:
        kw = dict(n_scales=2, n_octaves=6, max_points=1 << 20, profile=1, upsample=1)
    else:
        kw = dict(n_scales=2, n_octaves=6, max_points=1 << 20)
        img = np.rint(oracle_mod.upsample2x(img, w))  # integers, like a decoded image
        w, h = 2 * w, 2 * h
    pr = _params(sift, oracle_mod, **kw)
    ref, meta, total = oracle_mod.extract(pr, img, w)
    assert total == len(ref) > 20000
    try:
        got = _extract_gpu(sift, img, w)
    finally:
        sift.set_params()
    _assert_points_equal(oracle_mod, got, ref)
    sift.shutdown()


@pytest.mark.parametrize(
    "w,h,S,O,seed,up,qcap",
    [(640, 480, 2, 4, 1, 0, 65536), (333, 250, 3, 3, 2, 0, 65536), (400, 300, 2, 3, 3, 1, 65536),
     (257, 190, 3, 2, 4, 1, 65536), (640, 480, 2, 4, 5, 0, 0), (640, 480, 2, 4, 5, 0, 500)],
)
def test_profile_lowe_matches_oracle(sift, oracle_mod, w, h, S, O, seed, up, qcap):
    """Profile "lowe" (strict extrema, refinement, rejection; optionally an up-sampled first
    octave): no counterpart in the reference, so the oracle's definition is the only pin.  Bit
    for bit, through the queue path, the dense path (queue capacity 0) and a mix (500)."""
    kw = dict(n_scales=S, n_octaves=O, max_points=30000, profile=1, upsample=up)
    pr = _params(sift, oracle_mod, **kw)
    img = sift.synth_image(w, h, seed=seed, dense=True)
    ref, meta, total = oracle_mod.extract(pr, img, w)
    assert total == len(ref) > 200
    frac = np.abs(ref["xpos"] * (2.0 if up else 1.0) % 1.0)
    assert (frac > 0).mean() > 0.9, "positions are refined, not integers"
    sift.lib().sift_set_queue_cap(qcap)
    try:
        got = _extract_gpu(sift, img, w)
    finally:
        sift.lib().sift_set_queue_cap(65536)
        sift.set_params()
    _assert_points_equal(oracle_mod, got, ref)


def test_profile_lowe_batch_u8_and_quantized_descriptors(sift, oracle_mod):
    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=4, max_points=6000, profile=1, upsample=1)
    n, w, h = 3, 320, 240
    imgs = np.stack([sift.synth_image(w, h, seed=40 + i, dense=True) for i in range(n)])
    try:
        d_pts, counts, meta = sift.extract_batch(torch.from_numpy(imgs.astype(np.uint8)).cuda(), w, want_meta=True)
        q = sift.quantize_descriptors(d_pts, int(counts[0])).cpu().numpy()
    finally:
        sift.set_params()
    meta = meta.cpu().numpy().reshape(n, pr.max_points, 2)
    for i in range(n):
        ref, rmeta, total = oracle_mod.extract(pr, imgs[i], w)
        assert total == len(ref) == counts[i] > 100
        got = sift.points_to_numpy(d_pts, int(counts[i]), offset=i * pr.max_points)
        _assert_points_equal(oracle_mod, got, ref)
        # meta = (octave index in the processed pyramid, scale index the point ended on)
        key = lambda p, m: sorted(zip(p["scale"].tolist(), p["ypos"].tolist(), p["xpos"].tolist(), m[:, 0].tolist(), m[:, 1].tolist()))
        assert key(got, meta[i, : counts[i]]) == key(ref, rmeta)
    got0 = sift.points_to_numpy(d_pts, int(counts[0]))
    assert np.array_equal(q, oracle_mod.quantize_descriptors(got0))
    assert q.max() <= 255 and q.max() > 50


@pytest.mark.parametrize("w,h", [(19, 19), (30, 200), (200, 18), (65, 33), (129, 64)])
@pytest.mark.parametrize("profile", [0, 1])
def test_tiny_and_thin_images(sift, oracle_mod, w, h, profile):
    """Images smaller than the 19-tap support, one strip wide or a few rows tall: every pixel is
    apron.  Planes and (possibly empty) point lists still equal the oracle's."""
    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=3, max_points=2000, profile=profile,
                 extrema_thresh=2.0)
    img = sift.synth_image(w, h, seed=w * 1000 + h, dense=True)
    try:
        ref, _, total = oracle_mod.extract(pr, img, w)
        assert total == len(ref)
        got = _extract_gpu(sift, img, w)
    finally:
        sift.set_params()
    _assert_points_equal(oracle_mod, got, ref)


def test_full_size_1080p_three_scales(sift, oracle_mod):
    """1920x1080 with 3 scales and 5 octaves (the S = 3 kernels at full size), one dense image."""
    pr = _params(sift, oracle_mod, n_scales=3, n_octaves=5, max_points=20000)
    img = sift.synth_image(1920, 1080, seed=4242, dense=True)
    try:
        ref, _, total = oracle_mod.extract(pr, img, 1920)
        assert total == len(ref) > 2000
        d_pts, counts = sift.extract_batch(torch.from_numpy(img[None]).cuda(), 1920)
    finally:
        sift.set_params()
    _assert_points_equal(oracle_mod, sift.points_to_numpy(d_pts, int(counts[0])), ref)


def test_eight_slots_in_flight_with_changing_sizes(sift, oracle_mod):
    """All eight stream slots busy at once, each on its own stream, and the image size of a slot
    changes from round to round (its arena is re-planned while the other slots are in flight)."""
    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=3, max_points=2500)
    sizes = [(200, 150), (333, 200), (97, 180), (400, 120), (64, 64)]
    rng = np.random.default_rng(21)
    cache = {}
    for k, (w, h) in enumerate(sizes):
        img = sift.synth_image(w, h, seed=40 + k, dense=True)
        ref, _, total = oracle_mod.extract(pr, img, w)
        assert total == len(ref)
        cache[k] = (torch.from_numpy(img).cuda(), oracle_mod.sort_points(ref).tobytes(), len(ref))
    streams = [torch.cuda.Stream() for _ in range(8)]
    bufs = [sift.new_point_buffer() for _ in range(8)]
    sift.initDeviceConstant()
    for _ in range(10):
        sift.initDeviceVariables()
        pick = [int(rng.integers(0, len(sizes))) for _ in range(8)]
        for slot in range(8):
            w, h = sizes[pick[slot]]
            sift.extractSift(bufs[slot], cache[pick[slot]][0], w, w, h, streams[slot], slot, 0)
        for slot in range(8):
            n = sift.getPointCount(slot)
            assert n == cache[pick[slot]][2]
            got = sift.points_to_numpy(bufs[slot], n).view(oracle_mod.POINT_DTYPE)
            assert oracle_mod.sort_points(got).tobytes() == cache[pick[slot]][1]


def test_two_host_threads_share_a_device(sift, oracle_mod):
    """Two host threads extract on one GPU at the same time, each on its own slot and stream
    (the library takes the device's lock per call); both get the oracle's points, many times."""
    import threading

    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=4, max_points=2500)
    sift.initDeviceConstant()
    sift.initDeviceVariables()
    jobs = []
    for slot, (w, h, seed) in enumerate([(640, 480, 11), (500, 375, 12)]):
        img = sift.synth_image(w, h, seed=seed, dense=True)
        ref, _, total = oracle_mod.extract(pr, img, w)
        assert total == len(ref) > 100
        jobs.append((slot, w, h, torch.from_numpy(img).cuda(), sift.new_point_buffer(), ref))
    errors = []

    def work(slot, w, h, d_img, pts, ref):
        try:
            stream = torch.cuda.Stream()
            for _ in range(20):
                sift.extractSift(pts, d_img, w, w, h, stream, slot, 0)
                n = sift.getPointCount(slot)
                # the slot's counter accumulates (cudaSiftH.cu:323-352): this call's points are the last len(ref)
                got = sift.points_to_numpy(pts, n)[-len(ref):] if n <= 2500 else None
                if n >= 2500:
                    break
                _assert_points_equal(oracle_mod, got, ref)
        except Exception as e:  # noqa: BLE001
            errors.append((slot, repr(e)))

    threads = [threading.Thread(target=work, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("profile", [0, 1])
@pytest.mark.parametrize("kind", ["flat", "checker", "noise", "steps", "huge_values", "ramp"])
def test_degenerate_content(sift, oracle_mod, kind, profile):
    """Images that are all of one kind: nothing to find (flat, ramp), a candidate at every pixel
    (checkerboard, full-range noise: every queue and list overflows), divisions by zero in the
    edge test (steps) and values far outside 0..255.  Same bits as the oracle, no error."""
    w, h = 300, 220
    rng = np.random.default_rng(99)
    yy, xx = np.mgrid[0:h, 0:w]
    img = {
        "flat": np.full((h, w), 77.0),
        "checker": 255.0 * (((xx // 3) + (yy // 3)) % 2),
        "noise": rng.integers(0, 256, (h, w)).astype(np.float64),
        "steps": 200.0 * ((xx // 40) % 2) + 30.0 * ((yy // 25) % 2),
        "huge_values": 1.0e6 * np.sin(xx / 7.0) * np.cos(yy / 5.0) - 3.0e5,
        "ramp": 0.5 * xx + 0.25 * yy,
    }[kind].astype(np.float32)
    kw = dict(n_scales=2, n_octaves=3, max_points=3000, profile=profile)
    pr = _params(sift, oracle_mod, **kw)
    ref, _, total = oracle_mod.extract(pr, img, w)
    try:
        got = _extract_gpu(sift, img, w)
    finally:
        sift.set_params()
    if kind in ("flat", "ramp"):
        assert len(ref) == 0
    if total <= 3000:
        _assert_points_equal(oracle_mod, got, ref)
    else:  # over the cap: a subset of what the oracle finds without a cap
        assert len(got) == 3000
        pr_all = oracle_mod.params(**dict(kw, max_points=200000))
        every, _, _ = oracle_mod.extract(pr_all, img, w)
        have = {r.tobytes() for r in every}
        assert all(r.tobytes() in have for r in got)


def test_async_batch_entry_is_ordered_on_the_callers_stream(sift, oracle_mod):
    """sift_extract_batch_async: two batches queued back to back on one stream, uploads queued
    in between, no host wait until the end; each equals the blocking entry's result."""
    sift.set_params(n_scales=2, n_octaves=4, max_points=1500)
    cap, n, w, h = 1500, 6, 320, 240
    host = [np.stack([sift.synth_image(w, h, seed=900 + 10 * b + i, dense=True) for i in range(n)]) for b in range(2)]
    want = []
    for b in range(2):
        pts, counts = sift.extract_batch(torch.from_numpy(host[b]).cuda(), w)
        want.append((pts.cpu().numpy().copy(), counts.copy()))
    stream = torch.cuda.Stream()
    pinned = [torch.from_numpy(x).pin_memory() for x in host]
    d_img = [torch.empty((n, h, w), dtype=torch.float32, device="cuda") for _ in range(2)]
    d_pts = [torch.full((n * cap, sift.POINT_FLOATS), float("nan"), dtype=torch.float32, device="cuda") for _ in range(2)]
    d_cnt = [torch.full((n,), -1, dtype=torch.int32, device="cuda") for _ in range(2)]
    with torch.cuda.stream(stream):
        for b in range(2):
            d_img[b].copy_(pinned[b], non_blocking=True)
            sift.extract_batch_async(d_img[b], w, d_pts[b], d_cnt[b], stream=stream)
    stream.synchronize()
    for b in range(2):
        counts = d_cnt[b].cpu().numpy()
        assert np.array_equal(counts, want[b][1])
        got = d_pts[b].cpu().numpy()
        for i in range(n):
            a = sift.points_to_numpy(torch.from_numpy(got), int(counts[i]), offset=i * cap)
            r = sift.points_to_numpy(torch.from_numpy(want[b][0]), int(counts[i]), offset=i * cap)
            assert oracle_mod.sort_points(a).tobytes() == oracle_mod.sort_points(r).tobytes()
    sift.set_params()


def test_async_batches_in_flight_on_two_streams(sift, oracle_mod):
    """Non-blocking batch calls take the library's private streams and arenas in turn, so calls
    queued on different caller streams overlap on the device.  Five calls in flight (different
    images, two sizes, launch sets of 4 so that a call of 6 images spans both lanes), no host
    wait in between: each must equal the blocking entry's result for its images."""
    sift.set_params(n_scales=2, n_octaves=4, max_points=1200)
    sift.lib().sift_set_batch_chunk(4)
    try:
        cap = 1200
        shapes = [(6, 320, 240), (3, 200, 264), (6, 320, 240), (4, 200, 264), (5, 320, 240)]
        host = [np.stack([sift.synth_image(w, h, seed=7000 + 20 * b + i, dense=True) for i in range(n)])
                for b, (n, w, h) in enumerate(shapes)]
        want = []
        for b, (n, w, h) in enumerate(shapes):
            pts, counts = sift.extract_batch(torch.from_numpy(host[b]).cuda(), w)
            want.append((pts.cpu().numpy().copy(), counts.copy()))
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        d_img = [torch.from_numpy(x).cuda() for x in host]
        d_pts = [torch.full((n * cap, sift.POINT_FLOATS), float("nan"), dtype=torch.float32, device="cuda")
                 for n, _, _ in shapes]
        d_cnt = [torch.full((n,), -1, dtype=torch.int32, device="cuda") for n, _, _ in shapes]
        torch.cuda.synchronize()
        for b, (n, w, h) in enumerate(shapes):
            sift.extract_batch_async(d_img[b], w, d_pts[b], d_cnt[b], stream=streams[b & 1])
        torch.cuda.synchronize()
        for b, (n, w, h) in enumerate(shapes):
            counts = d_cnt[b].cpu().numpy()
            assert np.array_equal(counts, want[b][1]), b
            got = d_pts[b].cpu().numpy()
            for i in range(n):
                a = sift.points_to_numpy(torch.from_numpy(got), int(counts[i]), offset=i * cap)
                r = sift.points_to_numpy(torch.from_numpy(want[b][0]), int(counts[i]), offset=i * cap)
                assert oracle_mod.sort_points(a).tobytes() == oracle_mod.sort_points(r).tobytes(), (b, i)
    finally:
        sift.lib().sift_set_batch_chunk(256)
        sift.set_params()


def test_parameter_validation(sift):
    with pytest.raises(sift.SiftError):
        sift.set_params(profile=2)
    with pytest.raises(sift.SiftError):
        sift.set_params(profile=0, upsample=1)
    with pytest.raises(sift.SiftError):
        sift.set_params(n_scales=4)
    sift.set_params()
    assert sift.get_params().profile == 0
    d = torch.zeros((2, 40, 48), dtype=torch.float64, device="cuda")
    with pytest.raises(sift.SiftError):
        sift.extract_batch(d, 48)


def test_random_cases_slice():
    """A slice of tools/fuzz_parity.py (random sizes, parameters, profiles, queue capacities, caps,
    image kinds and entry points against the oracle), run as one separate process."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "80", "12345"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "80 cases, 0 bad" in r.stdout


def test_command_line_front_end(sift, oracle_mod, tmp_path):
    """sift_cli <list file> <destination> (the counterpart of Sift/src/main.cc): run as a
    separate process, its text files equal the oracle's."""
    import subprocess
    from pnm import write_pnm

    exe = os.path.join(os.path.dirname(os.path.abspath(sift.__file__)), "sift_cli")
    assert os.path.exists(exe), "make -C sift_amd/csrc builds it"
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    dst.mkdir()
    imgs = []
    for i in range(3):
        img = sift.synth_image(200 + 30 * i, 150, seed=810 + i, dense=True)
        write_pnm(str(src / f"im{i}.pgm"), img.astype(np.uint8))
        imgs.append(img)
    lst = tmp_path / "list.txt"
    lst.write_text("".join(f"{src / f'im{i}.pgm'}\n" for i in range(3)))
    r = subprocess.run([exe, str(lst), str(dst), "--octaves", "3", "--cap", "4000"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    pr = oracle_mod.params(n_scales=2, n_octaves=3, max_points=4000)
    for i, img in enumerate(imgs):
        ref, _, total = oracle_mod.extract(pr, img, img.shape[1])
        assert total == len(ref) > 10
        got = open(dst / f"im{i}.pgm.sift.txt").read().splitlines()
        assert sorted(got) == sorted(oracle_mod.format_points(ref).splitlines())
    r = subprocess.run([exe, str(tmp_path / "missing.txt"), str(dst)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "Unable to open file" in r.stderr


def test_slot_outlives_the_callers_stream(sift, oracle_mod):
    """The reference's caller creates a stream per image and destroys it after the read-back
    (Sift/src/sift.cu:96,130).  The slot must not hold on to it: the next image on the same slot
    has another size (the arena is planned afresh, which waits for the old work) and another
    stream, and getPointCount follows a stream that no longer exists either."""
    import ctypes as C

    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=3, max_points=20000)
    L = sift.lib()
    slot = 5
    sift.initDeviceVariables()
    total = 0
    for k, (w, h) in enumerate([(256, 192), (320, 200), (200, 264)]):
        img = sift.synth_image(w, h, seed=40 + k, dense=True)
        ref, _, _ = oracle_mod.extract(pr, img, w)
        d = torch.from_numpy(img).cuda()
        pts = sift.new_point_buffer()
        torch.cuda.synchronize()
        s = C.c_void_p()
        assert L.sift_stream_create(C.byref(s)) == 0
        if k:  # appends behind the earlier images' count: start this image's list at zero
            sift.initDeviceVariables()
        L.extractSift(C.c_void_p(pts.data_ptr()), C.c_void_p(d.data_ptr()), w, w, h, s, slot, 0)
        assert sift.last_error() == ""
        assert L.sift_stream_sync(s) == 0
        assert L.sift_stream_destroy(s) == 0
        n = sift.getPointCount(slot)
        assert n == len(ref) > 0
        _assert_points_equal(oracle_mod, sift.points_to_numpy(pts, n), ref)
        total += n
    acc, _ = sift.group_counts(slot)
    assert sum(acc) == n


def test_list_pipeline_on_several_workers_matches_the_pairwise_pipeline(sift, oracle_mod, tmp_path):
    """sift_extract_list_multi (image i on worker i mod n, one host thread per worker, decode /
    GPU / writer overlapped) against sift() (the reference's pairwise loop, Sift/src/sift.cu:64-143):
    the same files byte for byte, with 1 worker and with 2 and 3 workers sharing this box's GPU;
    grey and colour files, two image sizes mixed in one list."""
    import filecmp

    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=4, max_points=2500)
    paths, ref_counts = [], []
    for i in range(23):
        w, h = (320, 200) if i % 5 != 3 else (256, 192)  # every fifth image has another size than its batch
        g = sift.synth_image(w, h, seed=300 + i, dense=True).astype(np.uint8)
        p = tmp_path / f"im{i:02d}.{'ppm' if i % 4 == 1 else 'pgm'}"
        if i % 4 == 1:  # colour file whose grey value (Q14 weights) is g
            rgb = np.repeat(g[:, :, None], 3, axis=2)
            p.write_bytes(b"P6\n%d %d\n255\n" % (w, h) + rgb.tobytes())
        else:
            p.write_bytes(b"P5\n# comment\n%d %d\n255\n" % (w, h) + g.tobytes())
        paths.append(str(p))
        ref_counts.append(len(oracle_mod.extract(pr, g.astype(np.float32), w)[0]))
    import ctypes as C

    base = tmp_path / "pairwise"
    base.mkdir()
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    assert sift.lib().sift(str(base).encode(), arr, len(paths)) == 0, sift.last_error()
    for workers in (1, 2, 3):
        out = tmp_path / f"multi{workers}"
        out.mkdir()
        counts = sift.extract_list_multi(paths, str(out), workers)
        assert counts.tolist() == ref_counts
        names = sorted(os.listdir(base))
        assert names == sorted(os.listdir(out)) and len(names) == len(paths)
        for f in names:
            assert filecmp.cmp(base / f, out / f, shallow=False), f"{f} differs with {workers} workers"
    # the same with the library's default rule for small launch sets (batches of these small images
    # fall under the pixel limit: dense detector alone, one blur plane per wave)
    sift.lib().sift_set_dense_route_pixels(DENSE_ROUTE_DEFAULT)
    try:
        out = tmp_path / "multi_dense_route"
        out.mkdir()
        assert sift.extract_list_multi(paths, str(out), 2).tolist() == ref_counts
        for f in sorted(os.listdir(base)):
            assert filecmp.cmp(base / f, out / f, shallow=False), f"{f} differs on the dense route"
    finally:
        sift.lib().sift_set_dense_route_pixels(0)
    # without a destination only the counts come back
    assert sift.extract_list_multi(paths[:5], None, 2).tolist() == ref_counts[:5]
    with pytest.raises(sift.SiftError, match="missing.pgm"):  # the worker thread's message reaches the caller
        sift.extract_list_multi(paths[:2] + [str(tmp_path / "missing.pgm")], None, 1)
    assert sift.last_error() == ""


def test_small_jobs_take_the_dense_route(sift, oracle_mod):
    """A launch set of at most sift_set_dense_route_pixels() first-octave pixels (default: one
    1080p image) skips the candidate queues and runs the dense detector alone; larger sets queue.
    Seen through the queue counters of the slot; the points equal the oracle's either way, also
    for a batch cut so that some launch sets are small and some are not."""
    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=4, max_points=20000)
    w, h = 640, 480
    img = sift.synth_image(w, h, seed=77, dense=True)
    ref, _, _ = oracle_mod.extract(pr, img, w)
    d = torch.from_numpy(img).cuda()
    L = sift.lib()
    try:
        for route, queues in ((DENSE_ROUTE_DEFAULT, False), (w * h - 1, True), (w * h, False), (0, True)):
            L.sift_set_dense_route_pixels(route)
            pts = sift.new_point_buffer()
            sift.initDeviceVariables()
            sift.extractSift(pts, d, w, w, h, None, 0, 0)
            n = sift.getPointCount(0)
            _assert_points_equal(oracle_mod, sift.points_to_numpy(pts, n), ref)
            acc, queued = sift.group_counts(0)
            assert sum(acc) == n
            assert (sum(queued) > 0) == queues, f"route limit {route}: queued {queued}"
        # batch of 5 in launch sets of 2: sets of 2 images (over the limit below) and a last set of 1 (under it)
        imgs = np.stack([sift.synth_image(w, h, seed=80 + i, dense=True) for i in range(5)])
        refs = [oracle_mod.extract(pr, im, w)[0] for im in imgs]
        L.sift_set_dense_route_pixels(w * h + 1)
        L.sift_set_batch_chunk(2)
        got, counts = sift.extract_batch(torch.from_numpy(imgs).cuda(), w)
        for i in range(5):
            _assert_points_equal(oracle_mod, sift.points_to_numpy(got, int(counts[i]), offset=i * pr.max_points), refs[i])
    finally:
        L.sift_set_dense_route_pixels(0)
        L.sift_set_batch_chunk(256)


def test_release_arenas_gives_the_memory_back(sift, oracle_mod):
    """sift_release_arenas frees the planes, queues and lists of the stream slots and batch lanes
    (the reference frees its planes inside every call, cudaSiftH.cu:58, 172-177; here they are
    kept between calls): the device's free memory rises by what a batch had taken, and the next
    calls plan afresh and give the same points."""
    pr = _params(sift, oracle_mod, n_scales=2, n_octaves=4, max_points=3000)
    w, h, n = 640, 480, 24
    imgs = np.stack([sift.synth_image(w, h, seed=500 + i, dense=True) for i in range(n)])
    d = torch.from_numpy(imgs).cuda()
    ref0 = oracle_mod.extract(pr, imgs[0], w)[0]
    assert sift.lib().sift_release_arenas() == 0
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    pts, counts = sift.extract_batch(d, w)
    first = [oracle_mod.sort_points(sift.points_to_numpy(pts, int(counts[i]), offset=i * pr.max_points)).tobytes()
             for i in range(n)]
    one = _extract_gpu(sift, imgs[0], w)
    _assert_points_equal(oracle_mod, one, ref0)
    torch.cuda.synchronize()
    held = free0 - torch.cuda.mem_get_info()[0]
    assert held > n * w * h * 4 * 5  # at least the first octave's five planes per image (plus what torch took for pts)
    del pts
    assert sift.lib().sift_release_arenas() == 0
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    assert torch.cuda.mem_get_info()[0] > free0 - held // 8
    assert sift.getPointCount(0) == len(ref0)  # the counters outlive the arenas
    pts, counts = sift.extract_batch(d, w)
    again = [oracle_mod.sort_points(sift.points_to_numpy(pts, int(counts[i]), offset=i * pr.max_points)).tobytes()
             for i in range(n)]
    assert again == first
    assert first[0] == oracle_mod.sort_points(ref0).tobytes()
