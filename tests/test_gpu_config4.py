"""BASELINE configs[3] on the GPU: 4096 synthetic 1920x1080 images (seeds 1000..5095), 2 scales,
5 octaves, cap 2500, image i on GPU i mod G.  The reference treats images independently
(Sift/src/sift.cu:68-134 walks the list two at a time; only the per-slot counters are shared),
so the job shards by image index with no collective and a host-side gather.

On the one GPU of a test box the same partition and gather run at world size 1 and at world
size 2 with both ranks on device 0, and bench.py is started bare with --gpus 2 the way the
driver starts it.  Every result is held to the CPU oracle.
"""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

W, H, S, O, CAP = 1920, 1080, 2, 5, 2500
SEED0, N_IMAGES, CHUNK = 1000, 4096, 256
SAMPLED = list(range(0, N_IMAGES, 128)) + [1, N_IMAGES - 1]  # 34 images spread over the job, both ranks' shares


def _say(capsys, msg):
    with capsys.disabled():
        print(f"[config4 {time.strftime('%H:%M:%S')}] {msg}", flush=True)


def _bench(args, env_extra):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    return r


def test_bench_started_bare_with_two_gpus_runs_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no launcher: rc 0, one JSON line, two ranks, the gathered
    per-image counts equal to a one-rank run of the same job."""
    common = ["--mode", "config4", "--images", "64", "--batch", "16", "--steps", "1", "--warmup", "1",
              "--no-cpu-baseline", "--no-ceiling"]
    two = _bench(["--gpus", "2"] + common, {"BENCH_ONE_DEVICE": "1"})
    assert two.returncode == 0, two.stderr[-2000:]
    lines = [ln for ln in two.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, two.stdout
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "strong"
    assert d2["gathered_counts"]["images"] == 64

    one = _bench(["--gpus", "1"] + common, {})
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert d1["n_gpus"] == 1 and d1["gathered_counts"]["images"] == 64
    assert d1["gathered_counts"] == d2["gathered_counts"]
    assert d1["gathered_counts"]["keypoints"] > 0


def test_bench_reports_failure_of_a_rank():
    """A rank that cannot start (impossible arguments) makes the bare multi-GPU start exit non-zero."""
    r = _bench(["--gpus", "2", "--scales", "7", "--steps", "1", "--warmup", "0", "--batch", "2", "--no-cpu-baseline",
                "--no-ceiling", "--no-photo"], {"BENCH_ONE_DEVICE": "1"})
    assert r.returncode != 0


WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
import sift_amd
from sift_amd.multi_gpu import extract_sharded, gather_in_image_order, shard_indices
sys.path.insert(0, os.path.join({root!r}, "tests"))
from test_gpu_config4 import W, H, S, O, CAP, SEED0, N_IMAGES, SAMPLED, gpu_extract_fn, LazyImages

rank, world = int(sys.argv[1]), int(sys.argv[2])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[3], rank=rank, world_size=world)
torch.cuda.set_device(0)  # both ranks on the one GPU of the box
sift_amd.set_params(n_octaves=O, n_scales=S, max_points=CAP)
local = extract_sharded(LazyImages(N_IMAGES), W, gpu_extract_fn(), rank, world)
assert sorted(local) == shard_indices(N_IMAGES, rank, world)
out = gather_in_image_order(local, N_IMAGES, dist)
if rank == 0:
    counts = np.array([len(o) for o in out], np.int32)
    np.savez(sys.argv[4], counts=counts, **{{f"p{{i}}": out[i] for i in SAMPLED}})
    print("GATHERED", int(counts.sum()))
else:
    assert out is None
dist.barrier()
dist.destroy_process_group()
"""


class LazyImages:
    """The job's image list without holding it: indexing with a list of image numbers makes those
    images (8-bit, the synthetic generator's integers) on the spot."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        return make_images(list(idx))


def make_images(idx, threads=None):
    from concurrent.futures import ThreadPoolExecutor

    import sift_amd

    threads = threads or min(16, len(os.sched_getaffinity(0)))
    out = np.empty((len(idx), H, W), np.uint8)

    def some(lo):
        buf = np.empty((H, W), np.float32)
        for k in range(lo, min(lo + 8, len(idx))):
            sift_amd.lib().sift_synth_image(buf.ctypes.data, W, W, H, SEED0 + idx[k], 0)
            out[k] = buf  # the generator rounds to integers in [0, 255]

    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(some, range(0, len(idx), 8)))
    return out


def gpu_extract_fn():
    """extract_fn for multi_gpu.extract_sharded: 8-bit images through sift_extract_batch_u8 in launch
    sets of 256; returns host point records and counts."""
    import torch

    import sift_amd

    def fn(images, width):
        n = len(images)
        pts = np.zeros((n, CAP), sift_amd.POINT_DTYPE)
        counts = np.zeros(n, np.int32)
        d_pts = torch.empty((CHUNK * CAP, sift_amd.POINT_FLOATS), dtype=torch.float32, device="cuda")
        for lo in range(0, n, CHUNK):
            hi = min(n, lo + CHUNK)
            d = torch.from_numpy(np.ascontiguousarray(images[lo:hi])).cuda()
            _, c = sift_amd.extract_batch(d, width, d_pts=d_pts)
            counts[lo:hi] = c
            raw = d_pts.view(CHUNK, CAP, sift_amd.POINT_FLOATS)[:hi - lo, :int(c.max()) if len(c) else 0].cpu().numpy()
            for k in range(hi - lo):
                pts[lo + k, :c[k]] = np.ascontiguousarray(raw[k, :c[k]]).view(sift_amd.POINT_DTYPE).reshape(-1)
        return pts, counts

    return fn


def test_config4_all_4096_images_sharded_and_gathered(sift, oracle_mod, tmp_path, capsys):
    """The whole job: every image's keypoint count against the oracle, 34 images spread over the
    job point for point, at world size 1 (this process) and world size 2 (two processes, gloo,
    one device); the two gathers must be identical."""
    import torch

    from sift_amd.multi_gpu import extract_sharded, gather_in_image_order

    kw = dict(n_octaves=O, n_scales=S, max_points=CAP)
    sift.set_params(**kw)
    pr = oracle_mod.params(**kw)
    threads = min(16, len(os.sched_getaffinity(0)))

    # world size 2 first, in the background, while this process works through the oracle
    port = str(29500 + os.getpid() % 2000)
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    out2 = tmp_path / "world2.npz"

    # world size 1 and the oracle, chunk by chunk (the job as 8-bit images is 8.5 GB; a chunk is 0.5 GB)
    fn = gpu_extract_fn()
    gpu_counts = np.zeros(N_IMAGES, np.int32)
    cpu_counts = np.zeros(N_IMAGES, np.int32)
    gpu_pts, cpu_pts = {}, {}
    t0 = time.time()
    for lo in range(0, N_IMAGES, CHUNK):
        idx = list(range(lo, lo + CHUNK))
        imgs = make_images(idx, threads)
        local = extract_sharded(imgs, W, fn, 0, 1)
        part = gather_in_image_order(local, CHUNK, None)
        ref_pts, ref_counts = oracle_mod.extract_batch(pr, imgs.astype(np.float32), W, n_threads=threads)
        for k, i in enumerate(idx):
            gpu_counts[i], cpu_counts[i] = len(part[k]), ref_counts[k]
            if i in SAMPLED:
                gpu_pts[i], cpu_pts[i] = part[k], np.array(ref_pts[k][:ref_counts[k]])
        _say(capsys, f"images {lo}..{lo + CHUNK - 1}: {int(gpu_counts[lo:lo + CHUNK].sum())} keypoints, "
                     f"{time.time() - t0:.0f} s")
    assert cpu_counts.max() < CAP, "parity is defined below the cap"
    assert np.array_equal(gpu_counts, cpu_counts)
    assert gpu_counts.sum() > 100 * N_IMAGES
    for i in SAMPLED:
        a = oracle_mod.sort_points(gpu_pts[i].view(oracle_mod.POINT_DTYPE))
        b = oracle_mod.sort_points(cpu_pts[i].view(oracle_mod.POINT_DTYPE))
        assert a.tobytes() == b.tobytes(), f"image {i}"
    torch.cuda.synchronize()

    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port, str(out2)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHERED" in outs[0]
    _say(capsys, f"two ranks on one device gathered, {time.time() - t0:.0f} s")
    w2 = np.load(out2)
    assert np.array_equal(w2["counts"], cpu_counts)
    for i in SAMPLED:
        a = oracle_mod.sort_points(np.ascontiguousarray(w2[f"p{i}"]).view(oracle_mod.POINT_DTYPE).reshape(-1))
        b = oracle_mod.sort_points(cpu_pts[i].view(oracle_mod.POINT_DTYPE))
        assert a.tobytes() == b.tobytes(), f"image {i} (world size 2)"
