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


def _extract_gpu(sift, img_host, w):
    h, pitch = img_host.shape
    d_img = torch.from_numpy(np.nan_to_num(img_host, nan=-1e30)).cuda()
    pts = sift.new_point_buffer()
    sift.initDeviceConstant()
    sift.initDeviceVariables()
    sift.extractSift(pts, d_img, w, pitch, h, None, 0, 0)
    n = sift.getPointCount(0)
    return sift.points_to_numpy(pts, n)


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
