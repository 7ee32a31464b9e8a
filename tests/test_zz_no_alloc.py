"""The entry point that allocates nothing (sift_extract_noalloc) against the batch entry
it is a variant of (sift_extract_batch, one image): same points as sets, bit for bit, on two
calls in a row, the second with a smaller image (the static scratch keeps the first call's
contents), then a size the static buffer cannot hold."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _noalloc(sift, img, w):
    h, pitch = img.shape
    d_img = torch.from_numpy(img).cuda()
    cap = sift.get_params().max_points
    d_pts = torch.full((cap, sift.POINT_FLOATS), float("nan"), dtype=torch.float32, device="cuda")
    d_count = torch.full((1,), 12345, dtype=torch.int32, device="cuda")  # the call clears it itself
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    rc = sift.lib().sift_extract_noalloc(C.c_void_p(d_pts.data_ptr()), C.c_void_p(d_count.data_ptr()),
                                         C.c_void_p(d_img.data_ptr()), w, pitch, h, C.c_void_p(stream.cuda_stream))
    assert rc == 0, sift.last_error()
    stream.synchronize()
    n = int(d_count.item())
    return sift.points_to_numpy(d_pts, n)


def _batch_entry(sift, img, w):
    d_pts, counts = sift.extract_batch(torch.from_numpy(img[None]).cuda(), w)
    return sift.points_to_numpy(d_pts, int(counts[0]))


def _same(oracle_mod, a, b):
    assert len(a) == len(b) > 0, f"{len(a)} points against {len(b)}"
    a = oracle_mod.sort_points(a.view(oracle_mod.POINT_DTYPE))
    b = oracle_mod.sort_points(b.view(oracle_mod.POINT_DTYPE))
    assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("dense_route_pixels", [0, 2200000])
def test_no_alloc_entry_matches_the_batch_entry_twice_in_a_row(sift, oracle_mod, dense_route_pixels):
    """Both ways a one-image job can be detected: through the candidate queues (limit 0) and, the
    library's default for a job this small, by the dense detector alone behind the plane-per-wave
    pyramid kernel; the batch entry it is compared with always takes the queues."""
    sift.set_params(n_scales=2, n_octaves=4, max_points=2500)
    big = sift.synth_image(800, 640, seed=501, pitch=832)          # the reference's own image size, pitched
    small = sift.synth_image(333, 200, seed=502, dense=True)       # smaller, ragged
    try:
        sift.lib().sift_set_dense_route_pixels(dense_route_pixels)
        got_big = _noalloc(sift, big, 800)
        got_small = _noalloc(sift, small, 333)                      # second call: stale scratch around it
    finally:
        sift.lib().sift_set_dense_route_pixels(0)
    _same(oracle_mod, got_big, _batch_entry(sift, np.ascontiguousarray(big), 800))
    _same(oracle_mod, got_small, _batch_entry(sift, small, 333))
    # and the oracle agrees with both
    ref, _, _ = oracle_mod.extract(oracle_mod.params(n_scales=2, n_octaves=4, max_points=2500), small, 333)
    _same(oracle_mod, got_small, ref)


def test_no_alloc_entry_three_scales_and_the_lowe_profile(sift, oracle_mod):
    kw = dict(n_scales=3, n_octaves=4, max_points=4000, profile=1)
    sift.set_params(**kw)
    try:
        img = sift.synth_image(640, 480, seed=503, dense=True)
        _same(oracle_mod, _noalloc(sift, img, 640), _batch_entry(sift, img, 640))
    finally:
        sift.set_params()


def test_no_alloc_entry_rejects_what_the_static_buffer_cannot_hold(sift):
    sift.set_params(n_scales=2, n_octaves=5, max_points=2500)
    w, h = 1920, 1080
    d_img = torch.zeros((h, w), dtype=torch.float32, device="cuda")
    d_pts = torch.zeros((2500, sift.POINT_FLOATS), dtype=torch.float32, device="cuda")
    d_count = torch.zeros((1,), dtype=torch.int32, device="cuda")
    rc = sift.lib().sift_extract_noalloc(C.c_void_p(d_pts.data_ptr()), C.c_void_p(d_count.data_ptr()),
                                         C.c_void_p(d_img.data_ptr()), w, w, h, None)
    msg = sift.last_error()
    sift.lib().sift_clear_error()
    sift.set_params()
    assert rc == -1 and "static buffer holds" in msg and str(sift.lib().sift_noalloc_scratch_bytes()) in msg


def test_no_alloc_entry_is_never_timed(sift, oracle_mod):
    """Per-kernel timing records events, and creating an event allocates: with timing switched
    on the entry still launches untimed (the timers see no launch) and gives the same points."""
    sift.set_params(n_scales=2, n_octaves=3, max_points=2500)
    img = sift.synth_image(320, 240, seed=504, dense=True)
    want = _batch_entry(sift, img, 320)
    sift.timing_reset()
    sift.timing_enable(True)
    try:
        got = _noalloc(sift, img, 320)
        launches = [sift.timing_read(k)[1] for k in (sift.K_PYRAMID, sift.K_VERIFY, sift.K_DETECT, sift.K_KEYPOINT)]
    finally:
        sift.timing_enable(False)
    assert launches == [0, 0, 0, 0]
    _same(oracle_mod, got, want)
    sift.set_params()


def test_no_alloc_entry_runs_under_stream_capture_and_leaves_free_memory_alone(sift, oracle_mod):
    """The entry's defining property.  After one warm-up call (which looks up the static buffer's
    address) the call is recorded into a HIP graph: under capture an allocation, an event created
    for timing or a wait on the stream fails the capture.  The graph is replayed and its points are
    held to the batch entry's; the device's free memory is the same before and after the calls."""
    sift.set_params(n_scales=2, n_octaves=4, max_points=2500)
    w, h = 640, 480
    img = sift.synth_image(w, h, seed=505, dense=True)
    want = _batch_entry(sift, img, w)
    d_img = torch.from_numpy(img).cuda()
    d_pts = torch.full((2500, sift.POINT_FLOATS), float("nan"), dtype=torch.float32, device="cuda")
    d_count = torch.full((1,), 777, dtype=torch.int32, device="cuda")
    args = (C.c_void_p(d_pts.data_ptr()), C.c_void_p(d_count.data_ptr()), C.c_void_p(d_img.data_ptr()), w, w, h)
    stream = torch.cuda.Stream()
    assert sift.lib().sift_extract_noalloc(*args, C.c_void_p(stream.cuda_stream)) == 0, sift.last_error()  # warm-up
    stream.synchronize()
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        assert sift.lib().sift_extract_noalloc(*args, C.c_void_p(stream.cuda_stream)) == 0, sift.last_error()
    stream.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before, "the entry changed the device's free memory"

    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        rc = sift.lib().sift_extract_noalloc(*args, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, sift.last_error()
    d_pts.fill_(float("nan"))
    d_count.fill_(777)
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    n = int(d_count.item())
    _same(oracle_mod, sift.points_to_numpy(d_pts, n), want)
    graph.replay()  # and again: the scratch keeps the last replay's contents
    torch.cuda.synchronize()
    _same(oracle_mod, sift.points_to_numpy(d_pts, int(d_count.item())), want)
    sift.set_params()
