"""CPU suite for the shipped library: the C ABI loads and exports what the header
declares, and the host-side logic (tables, parameters, loader, writer, sharding)
is right.  No kernel runs here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from pnm import read_pnm, write_pnm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/Sift"


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def declared_functions():
    src = open(os.path.join(ROOT, "include", "sift_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", src)))


def test_library_exports_every_declared_symbol(sift):
    names = declared_functions()
    assert {"initDeviceConstant", "initDeviceVariables", "extractSift", "getPointCount",
            "blurOctave", "sift_extract_batch"} <= set(names)
    raw = ctypes.CDLL(sift.LIB_PATH)
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing
    assert set(names) == set(sift.SYMBOLS), "the Python binding covers exactly the header"


def test_point_record_layout(sift):
    assert sift.POINT_DTYPE.itemsize == 528  # Sift/src/sift.h:7-13
    assert sift.POINT_DTYPE.fields["data"][1] == 16


def test_missing_library_fails_loudly(sift, monkeypatch):
    monkeypatch.setattr(sift, "_lib", None)
    monkeypatch.setattr(sift, "LIB_PATH", "/nonexistent/libsift_amd.so")
    with pytest.raises(sift.SiftError, match="no CPU fallback"):
        sift.lib()


def test_params_roundtrip_and_validation(sift):
    d = sift.default_params()
    assert (d.n_octaves, d.n_scales, d.max_points) == (4, 2, 2500)  # utils.h:7-10
    assert (d.sigma, d.r_thresh, d.extrema_thresh) == (np.float32(1.6), 10.0, 10.0)
    sift.set_params(n_octaves=5, n_scales=3, max_points=7)
    g = sift.get_params()
    assert (g.n_octaves, g.n_scales, g.max_points) == (5, 3, 7)
    for bad in (dict(n_scales=5), dict(n_octaves=0), dict(n_octaves=9), dict(max_points=0), dict(sigma=-1.0)):
        with pytest.raises(sift.SiftError):
            sift.set_params(**bad)
    assert sift.get_params().n_scales == 3, "a rejected call changes nothing"
    sift.set_params()


@pytest.mark.parametrize("S", [2, 3])
def test_library_tables_equal_oracle_tables(sift, oracle_mod, S):
    sift.set_params(n_scales=S)
    got = sift.get_tables()
    want = oracle_mod.tables(oracle_mod.params(n_scales=S))
    for g, w in zip(got, want):
        assert np.array_equal(bits(g), bits(w))
    sift.set_params()


def test_text_writer_equals_oracle_writer(sift, oracle_mod):
    rng = np.random.default_rng(3)
    pts = np.zeros(5, sift.POINT_DTYPE)
    pts["xpos"] = rng.integers(9, 900, 5)
    pts["ypos"] = rng.integers(9, 900, 5)
    pts["scale"] = [1.6, 2.2627418, 3.2, 4.5254836, 6.4]
    pts["orientation"] = 11.25 * rng.integers(0, 32, 5)
    pts["data"] = rng.uniform(0, 0.2, (5, 128))
    assert sift.format_points(pts) == oracle_mod.format_points(pts.view(oracle_mod.POINT_DTYPE))


def test_fast_decimal_formatting_is_printf(sift, oracle_mod):
    """The writer formats the 128 bins of a point with integer arithmetic; the bytes must be those
    of printf("%.6f"), which the oracle's writer calls: exact ties (odd multiples of 2^-7 have seven
    decimals ending in 5), values next to them, denormals, random bit patterns, and the values
    that take the printf path (negative, 1000 and above, infinities, NaN)."""
    rng = np.random.default_rng(8)
    ties = np.arange(1, 4001, 2, dtype=np.float64) / 128.0  # up to 31.2
    special = np.array([0.0, -0.0, 1.0, 0.2, 0.5, 0.9999995, 0.99999949, 999.9999, 999.99994, 1000.0, 123456.0,
                        -0.25, 1e-45, 1e-38, 4.9999997e-7, 5.0000003e-7, 5e-7, 1.5e-6, 2.5e-6, np.inf, -np.inf,
                        np.nan, 3.4e38, 16777216.0], dtype=np.float32)
    near = np.concatenate([np.nextafter(ties.astype(np.float32), np.float32(0)),
                           np.nextafter(ties.astype(np.float32), np.float32(1e9))])
    bits_ = rng.integers(0, 2 ** 32, 40000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    small = rng.uniform(0, 1, 40000).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 3, 40000).astype(np.float32)
    vals = np.concatenate([ties.astype(np.float32), special, near, bits_, small])
    vals = np.concatenate([vals, np.zeros((-len(vals)) % 128, np.float32)])
    pts = np.zeros(len(vals) // 128, sift.POINT_DTYPE)
    pts["data"] = vals.reshape(-1, 128)
    got = sift.format_points(pts)
    want = oracle_mod.format_points(pts.view(oracle_mod.POINT_DTYPE))
    assert got == want


def test_pnm_loader_matches_gray_rule(sift, oracle_mod, tmp_path):
    rng = np.random.default_rng(4)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    write_pnm(tmp_path / "c.ppm", rgb)
    write_pnm(tmp_path / "g.pgm", rgb[..., 1])
    assert np.array_equal(sift.load_pnm_gray(tmp_path / "c.ppm"), oracle_mod.gray_q14(rgb).astype(np.float32))
    assert np.array_equal(sift.load_pnm_gray(tmp_path / "g.pgm"), rgb[..., 1].astype(np.float32))
    with pytest.raises(sift.SiftError):
        sift.load_pnm_gray(tmp_path / "absent.ppm")
    if os.path.isdir(REF):  # the reference's known-answer file
        res = read_pnm(os.path.join(REF, "result/result.ppm"))[..., 0]
        assert np.array_equal(sift.load_pnm_gray(os.path.join(REF, "img/img3.ppm")), res.astype(np.float32))


def test_synthetic_images_are_integer_valued_and_seeded(sift):
    a = sift.synth_image(320, 200, seed=5)
    b = sift.synth_image(320, 200, seed=5, pitch=384)
    c = sift.synth_image(320, 200, seed=6)
    assert a.min() >= 0 and a.max() <= 255 and np.array_equal(a, np.round(a))
    assert np.array_equal(a, b[:, :320]) and not np.array_equal(a, c)
    assert a.std() > 5


def test_shard_indices(sift):
    from sift_amd.multi_gpu import shard_indices

    for n in (0, 1, 7, 4096):
        for world in (1, 2, 4, 8):
            parts = [shard_indices(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)


WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch.distributed as dist
import oracle, sift_amd
from sift_amd.multi_gpu import extract_sharded, gather_in_image_order

dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
pr = oracle.params(n_octaves=2, max_points=4000)
imgs = np.stack([sift_amd.synth_image(160, 120, seed=40 + i, dense=True) for i in range(5)])
fn = lambda local, w: oracle.extract_batch(pr, local, w)   # stands in for the GPU extractor
local = extract_sharded(imgs, 160, fn, dist.get_rank(), 2)
assert sorted(local) == list(range(dist.get_rank(), 5, 2))
out = gather_in_image_order(local, 5, dist)
if dist.get_rank() == 0:
    ref_pts, ref_counts = oracle.extract_batch(pr, imgs, 160)
    assert [len(o) for o in out] == ref_counts.tolist() and sum(ref_counts) > 0
    for i in range(5):
        assert out[i].tobytes() == ref_pts[i][: ref_counts[i]].tobytes()
    print("GATHER_OK", ref_counts.tolist())
else:
    assert out is None
dist.barrier()
dist.destroy_process_group()
"""


def test_two_rank_sharding_and_gather_over_gloo(tmp_path):
    pytest.importorskip("torch")
    port = 29500 + os.getpid() % 2000
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0]


def test_draw_points_writes_a_ppm_with_marks(sift, oracle_mod, tmp_path):
    """sift_draw_points is host code: a grey image, two hand-made points, a PPM out."""
    C = ctypes
    w, h = 96, 64
    gray = np.full((h, w), 100, np.float32)
    pts = np.zeros(2, oracle_mod.POINT_DTYPE)
    pts["xpos"], pts["ypos"], pts["scale"], pts["orientation"] = [20, 30], [30, 10], [1.6, 3.2], [0, 90]
    scale = np.array([1.0, 2.0], np.float32)  # the second point lives on an octave-1 grid
    out = tmp_path / "marks.ppm"
    rc = sift.lib().sift_draw_points(os.fsencode(str(out)), gray.ctypes.data_as(C.POINTER(C.c_float)), w, w, h,
                                     pts.ctypes.data, 2, scale.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0
    img = read_pnm(str(out))
    assert img.shape == (h, w, 3)
    red = (img[..., 0] == 255) & (img[..., 1] == 40)
    green = (img[..., 1] == 255) & (img[..., 0] == 40)
    assert red.sum() > 40 and green.sum() > 8
    assert green[30, 21] and green[21, 60], "spokes start at the (scaled) positions and follow the orientation"
    ys, xs = np.nonzero(red)
    d0 = np.hypot(xs - 20, ys - 30)
    d1 = np.hypot(xs - 60, ys - 20)
    assert np.all((np.abs(d0 - 4.8) < 1.0) | (np.abs(d1 - 9.6) < 1.0)), "circles of radius 3*scale"
    untouched = ~(red | green)
    assert np.all(img[untouched] == 100)


def test_list_partition_is_image_mod_workers(sift):
    """Image i of a list goes to worker i mod n (BASELINE configs[3]); every image has exactly one worker."""
    L = sift.lib()
    for n in (1, 2, 4, 8):
        owners = [L.sift_list_partition(i, n) for i in range(4096)]
        assert owners == [i % n for i in range(4096)]
        assert sorted(set(owners)) == list(range(n))
    assert L.sift_list_partition(5, 0) == -1


def test_host_code_and_oracle_under_address_and_ub_sanitizers():
    """`make -C tests/asan run` (also `make -C oracle asan`, `make -C sift_amd/csrc asan`): the oracle
    and the C host code (sift_host.c, pipeline.c against a CPU stand-in of the HIP layer) built with
    -fsanitize=address,undefined and driven through arenas, slots, launch sets, loaders, the formatter
    and both file pipelines.  CPU only."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-s", "-C", os.path.join(root, "tests", "asan"), "run"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "oracle driver finished" in r.stdout and "host driver finished" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_host_code_under_thread_sanitizer():
    """`make -C tests/asan tsan`: the same host driver (two device threads, the list pipeline's
    workers and job pool, the per-device locks and the parameter lock) under -fsanitize=thread."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-s", "-C", os.path.join(root, "tests", "asan"), "tsan"], capture_output=True, text=True,
                       timeout=600)
    if "unexpected memory mapping" in r.stderr:  # (the sanitizer's runtime refuses some kernels' address-space layouts)
        pytest.skip("ThreadSanitizer cannot run on this machine")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "host driver finished" in r.stdout
    assert "WARNING: ThreadSanitizer" not in r.stderr


def test_the_integration_notes_binding_compiles_and_links(sift, tmp_path):
    """The C++ stub INTEGRATION.md gives a maintainer of Sift/src (a header in place of
    cudaSiftH.h, Sift/src/cudaSiftH.h:6-30) is taken from the notes as it stands and compiled
    together with a caller shaped like the reference's loop (Sift/src/sift.cu:94-130: one stream
    per queued image passed by reference, initDeviceConstant / initDeviceVariables every round,
    extractSift per image, getPointCount as the synchronisation point), then linked against the
    shipped library.  Nothing is run: no GPU here."""
    notes = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```cpp\n(// sift_amd_shim\.h.*?)```", notes, flags=re.S)
    assert m, "the shim header is missing from INTEGRATION.md"
    (tmp_path / "sift_amd_shim.h").write_text(m.group(1))
    (tmp_path / "caller.cc").write_text(r"""
#include "sift_amd_shim.h"
#include <cstdio>
static_assert(sizeof(SiftPoint) == 528, "Sift/src/sift.h:7-13");
int run_pair(float *d_img[2], SiftPoint *d_pts[2], int w, int pitch, int h)
{
    hipStream_t stream[2];
    for (int i = 0; i < 2; ++i) if (hipStreamCreate(&stream[i]) != hipSuccess) return -1;
    initDeviceConstant();
    initDeviceVariables();
    for (int i = 0; i < 2; ++i) extractSift(d_pts[i], d_img[i], w, pitch, h, stream[i], i, 0);
    int total = 0;
    for (int i = 0; i < 2; ++i) total += getPointCount(i) - 1;
    for (int i = 0; i < 2; ++i) hipStreamDestroy(stream[i]);
    if (*sift_last_error()) std::fprintf(stderr, "%s\n", sift_last_error());
    return total;
}
int main() { return 0; }
""")
    lib_dir = os.path.dirname(sift.LIB_PATH)
    cmd = ["g++", "-std=c++11", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
           "-I", "/opt/rocm/include", str(tmp_path / "caller.cc"), "-o", str(tmp_path / "caller"),
           "-L", lib_dir, "-lsift_amd", "-L", "/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_strip_heights_of_the_pyramid_launches(sift):
    """The launch shapes the measurements were made with: a batch that fills the chip many times over
    is cut into few tall strips (18 apron rows per strip are wasted work), a launch the chip holds in
    one round into strips priced by the waves that share a SIMD.  Host arithmetic only."""
    rows = sift.lib().sift_debug_strip_rows
    # BASELINE configs[2]: 256 images of 1920x1080, 2 scales, octave after octave
    assert [rows(w, h, 256, 2, 0) for (w, h) in ((1920, 1080), (960, 540), (480, 270), (240, 135))] == [540, 135, 90, 45]
    # one 1080p image: the first octave with one wave per tile, the later ones with one plane per wave
    assert rows(1920, 1080, 1, 2, 0) == 16
    assert 1 <= rows(120, 68, 1, 2, 0) <= 2
    for (w, h) in ((960, 540), (480, 270), (240, 135), (120, 68)):
        r = rows(w, h, 1, 2, 1)
        waves = -(-w // 64) * -(-h // r) * 5
        assert 1 <= r <= h and waves <= 8 * 1024, (w, h, r)
    assert rows(960, 540, 1, 2, 1) > rows(120, 68, 1, 2, 1)
    # any geometry gives strips inside the image
    rng = np.random.default_rng(5)
    for _ in range(300):
        w, h, n, s = int(rng.integers(1, 5000)), int(rng.integers(1, 3000)), int(rng.integers(1, 300)), int(rng.integers(2, 4))
        for kind in (0, 1, 2):
            assert 1 <= rows(w, h, n, s, kind) <= h
    assert rows(0, 10, 1, 2, 0) == -1 and rows(10, 10, 1, 4, 0) == -1 and rows(10, 10, 1, 2, 7) == -1
