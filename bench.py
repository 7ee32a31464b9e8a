#!/usr/bin/env python3
"""Headline benchmark: SIFT extraction throughput at 1080p on MI355X.

A step is one pass of the whole hot path (pyramid, detection, orientation,
descriptors) over one batch of 256 synthetic 1920x1080 float images, 2 scales,
5 octaves, already resident in HBM.  With N > 1 (launched by torchrun, one rank
per GPU) every rank owns its own batch: images are independent, there is no
collective on the data path, so scaling is weak and `value` is the sum over
ranks divided by the slowest rank's time.

`--mode config4` is BASELINE.json's configs[3]: 4096 images (seeds 1000..5095), image i
on rank i mod N, every rank's share resident in its HBM and worked through in batches of
256; `value` is 4096 divided by the slowest rank's time (strong scaling) and rank 0
gathers the per-image keypoint counts in image order.

Started bare with `--gpus N` for N > 1 (no WORLD_SIZE in the environment) the program
starts its own N ranks, one process per GPU, before anything in it has touched a GPU, relays
rank 0's line and exits non-zero if any rank failed.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, S, O, CAP = 1920, 1080, 2, 5, 2500
BATCH = 256
SEED0 = 1000
HBM_PEAK = 8.0e12  # B/s, MI355X spec


def octave_pixels(w, h, n_oct):
    out = []
    for _ in range(n_oct):
        out.append(w * h)
        w, h = (w + 1) // 2, (h + 1) // 2
    return out


def algorithmic_bytes(s):
    """Bytes per pixel per octave, float32 planes.

    `pyramid`: what the fused row+column kernel has to move: read the octave's
    source once, write the S+3 blur planes (DoG values are formed by their
    readers; the next octave reads blur plane S in place).
    `detect`: the dense detector reads the S+3 blur planes."""
    return {"pyramid": (1 + (s + 3)) * 4.0, "detect": (s + 3) * 4.0}


def source_hash():
    """Hash of the sources that decide what the pyramid kernel moves; tools/pmc_summary.py stores
    it beside the counted traffic so that a figure counted on other code is not reported."""
    h = hashlib.sha256()
    for rel in ("sift_amd/csrc/sift_kernels.hip", "sift_amd/csrc/sift_host.c"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def spawn_ranks(n):
    """One child process per GPU, this same command line, rendezvous on 127.0.0.1.  The parent has
    not imported torch or the library: nothing here has initialised a GPU.  Returns the exit code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    share = max(1, host_threads() // n)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS=str(share))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def host_threads():
    """Threads for host-side work: the CPUs this process may run on, within the share the
    job was given (OMP_NUM_THREADS carries it on the GPU boxes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        share = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        share = 0
    return max(1, min(n, share) if share > 0 else n)


def photo_density_rate(sift_amd, torch, steps=3, n=64):
    """Images/s on images with the keypoint density of a photograph: the committed 320x256 crop
    of the reference's Sift/img/img1.ppm, mirrored and tiled to 1920x1080 (about 11 k keypoints
    per image before the cap of 2500)."""
    path = os.path.join(ROOT, "tests", "golden", "real_img1_crop.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)["gray"].astype(np.float32)
    row = np.concatenate([g, g[:, ::-1]] * 3, axis=1)
    img = np.ascontiguousarray(np.concatenate([row, row[::-1]] * 3, axis=0)[:H, :W])
    batch = torch.from_numpy(np.stack([np.roll(img, 7 * i, axis=1) for i in range(n)])).cuda()
    d_pts = torch.empty((n * CAP, sift_amd.POINT_FLOATS), dtype=torch.float32, device="cuda")
    _, counts = sift_amd.extract_batch(batch, W, d_pts=d_pts)
    torch.cuda.synchronize()
    sift_amd.timing_reset()
    sift_amd.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        _, counts = sift_amd.extract_batch(batch, W, d_pts=d_pts)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sift_amd.timing_enable(False)
    kern = {name: sift_amd.timing_read(kid)[0] / (n * steps)
            for name, kid in (("pyramid", sift_amd.K_PYRAMID), ("verify", sift_amd.K_VERIFY),
                              ("detect", sift_amd.K_DETECT), ("keypoint", sift_amd.K_KEYPOINT))}
    return {"images_per_s": n * steps / dt, "keypoints_per_image": float(counts.mean()), "images": n,
            "kernels_ms_per_image": kern,
            "source": "crop of Sift/img/img1.ppm tiled to 1920x1080 (tests/golden/real_img1_crop.npz)"}


def streaming_rate(sift_amd, torch, d_imgs, steps):
    """Images/s when the same batches arrive as a stream: every step is one non-blocking batch call,
    steps alternate between two caller streams and two output buffers, and the host waits once at
    the end.  The library gives consecutive non-blocking calls its two private streams and arenas
    in turn, so one step's short tail (small octaves, verification, keypoints) runs beside the next
    step's first octave.  Not the headline: kernels of two steps share the chip, so per-kernel
    timings of this mode say nothing about a kernel."""
    n = d_imgs.shape[0]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    d_pts = [torch.empty((n * CAP, sift_amd.POINT_FLOATS), dtype=torch.float32, device="cuda") for _ in range(2)]
    d_cnt = torch.zeros((steps + 2, n), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for k in range(2):  # both arenas exist before the clock starts
        sift_amd.extract_batch_async(d_imgs, W, d_pts[k], d_cnt[k], stream=streams[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        sift_amd.extract_batch_async(d_imgs, W, d_pts[k & 1], d_cnt[2 + k], stream=streams[k & 1])
    h_cnt = d_cnt.cpu()  # (waits for both streams: the copy is queued behind a device synchronisation)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = bool((h_cnt == h_cnt[0]).all())
    return {"images_per_s": n * steps / dt, "steps": steps, "in_flight": 2, "counts_equal_every_step": same}


def gen_images(n, seed0, dense, threads):
    import sift_amd

    imgs = np.empty((n, H, W), np.float32)

    def one(i):
        sift_amd.lib().sift_synth_image(imgs[i].ctypes.data, W, W, H, seed0 + i, int(dense))

    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(one, range(n)))
    return imgs


def host_fed_rates(sift_amd, torch, host, d_pts, steps):
    """Images/s when every step has to bring its batch over PCIe from pinned host memory.
    The batch travels in four parts on a copy stream into one of two device buffers, and the
    upload of the next step's batch is queued before this step's parts are extracted, so in the
    steady state the link never waits for the extraction.  Not the headline: the boundary of the
    hot path takes device buffers."""
    res = {}
    n = host.shape[0]
    n_parts = 4 if n >= 64 else 1
    bounds = [n * k // n_parts for k in range(n_parts + 1)]
    for name, arr in (("u8", host.astype(np.uint8)), ("f32", host)):
        pinned = torch.from_numpy(arr).pin_memory()
        dev = [torch.empty_like(pinned, device="cuda") for _ in range(2)]
        copy = torch.cuda.Stream()
        done = [[torch.cuda.Event() for _ in range(n_parts)] for _ in range(2)]
        free = [torch.cuda.Event() for _ in range(2)]  # the buffer's last extraction has finished
        cap = d_pts.shape[0] // n

        def upload(buf):
            with torch.cuda.stream(copy):
                copy.wait_event(free[buf])
                for k in range(n_parts):
                    lo, hi = bounds[k], bounds[k + 1]
                    dev[buf][lo:hi].copy_(pinned[lo:hi], non_blocking=True)
                    done[buf][k].record(copy)

        def step(buf, more):
            if more:
                upload(1 - buf)
            for k in range(n_parts):
                lo, hi = bounds[k], bounds[k + 1]
                torch.cuda.current_stream().wait_event(done[buf][k])
                sift_amd.extract_batch(dev[buf][lo:hi], W, d_pts=d_pts[lo * cap:hi * cap])
            free[buf].record(torch.cuda.current_stream())

        for f in free:
            f.record(torch.cuda.current_stream())
        upload(0)
        step(0, True)  # warm-up; leaves the next batch on its way
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step((i + 1) % 2, True)  # K uploads and K extractions inside the timed region
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name] = {"images_per_s": n * steps / dt, "host_GBps": arr.nbytes * steps / dt / 1e9}
        del dev, pinned
    return res


def main():
    global S
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--streaming", action="store_true",
                    help="add the two-batches-in-flight side measurement (its kernels overlap: leave it out of profiled runs)")
    ap.add_argument("--chunk", type=int, default=0, help="images per launch set (0 = library default)")
    ap.add_argument("--lanes", type=int, default=0, help="launch sets in flight (0 = library default)")
    ap.add_argument("--strip-rows", type=int, default=0, help="rows per strip of the pyramid pass (0 = chosen per launch)")
    ap.add_argument("--columns", type=int, default=0, choices=[0, 1, 2],
                    help="columns per lane of the pyramid pass (0 = library default)")
    ap.add_argument("--columns-min-width", type=int, default=0, help="narrowest octave that takes two columns per lane")
    ap.add_argument("--dense", action="store_true", help="high keypoint density images")
    ap.add_argument("--queue-cap", type=int, default=-1,
                    help="candidate queue entries per image and scale (0 = dense detector only; -1 = library default)")
    ap.add_argument("--scales", type=int, default=S, help="scales per octave (2 = the headline configuration, or 3)")
    ap.add_argument("--u8", action="store_true", help="8-bit images in HBM, converted by the pyramid kernel's fetch")
    ap.add_argument("--host-fed", action="store_true",
                    help="also time the batch arriving from pinned host memory over PCIe (8-bit and float), "
                         "upload of one half overlapped with extraction of the other; reported beside, never as, value")
    ap.add_argument("--profile", choices=["ref", "lowe"], default="ref",
                    help="semantics profile (ref = the headline configuration)")
    ap.add_argument("--upsample", action="store_true", help="profile lowe: doubled first octave")
    ap.add_argument("--mode", choices=["config3", "config4"], default="config3",
                    help="config3: every rank its own batch (weak scaling, the headline); config4: 4096 images "
                         "dealt over the ranks, image i on rank i mod N (strong scaling)")
    ap.add_argument("--images", type=int, default=4096, help="config4: images in the whole job")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the device-copy ceiling measurement")
    ap.add_argument("--no-photo", action="store_true", help="skip the photograph-density extra")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=256)
    args = ap.parse_args()
    S = args.scales

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import sift_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torchrun", file=sys.stderr)
        sys.exit(2)
    # BENCH_ONE_DEVICE=1 puts every rank on device 0: a rehearsal of the N > 1 path on a one-GPU box
    torch.cuda.set_device(0 if os.environ.get("BENCH_ONE_DEVICE") else local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        # the data path has no collective; only the timing and the keypoint count are reduced,
        # as host scalars, so the rendezvous runs over gloo
        # (its transport announces itself on stdout, which must carry the JSON line only)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
            dist.barrier()
        finally:
            os.dup2(saved, 1)
            os.close(saved)

    sift_amd.lib()  # raises if the HIP library is missing
    profile = 1 if args.profile == "lowe" else 0
    sift_amd.set_params(n_octaves=O, n_scales=S, max_points=CAP, profile=profile, upsample=int(args.upsample))
    if args.chunk:
        sift_amd.lib().sift_set_batch_chunk(args.chunk)
    if args.lanes:
        sift_amd.lib().sift_set_batch_lanes(args.lanes)
    if args.strip_rows:
        sift_amd.lib().sift_set_strip_rows(args.strip_rows)
    if args.queue_cap >= 0:
        sift_amd.lib().sift_set_queue_cap(args.queue_cap)
    if args.columns:
        sift_amd.lib().sift_set_pyramid_columns(args.columns, args.columns_min_width)

    threads = host_threads()
    config4 = args.mode == "config4"
    if config4:
        # image i of the job lives on rank i mod world; all of a rank's images are resident
        mine = list(range(rank, args.images, world))
        n_local = len(mine)
        host = np.empty((n_local, H, W), np.float32)

        def one(k):
            sift_amd.lib().sift_synth_image(host[k].ctypes.data, W, W, H, SEED0 + mine[k], int(args.dense))

        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(one, range(n_local)))
    else:
        n_local = args.batch
        host = gen_images(args.batch, SEED0 + rank * args.batch, args.dense, threads)
    d_imgs = torch.from_numpy(host.astype(np.uint8) if args.u8 else host).cuda()  # the images hold integers
    d_pts = torch.empty((n_local * CAP, sift_amd.POINT_FLOATS), dtype=torch.float32, device="cuda")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        """One pass of the hot path over this rank's images, a batch at a time."""
        got = []
        for lo in range(0, n_local, args.batch):
            hi = min(n_local, lo + args.batch)
            _, c = sift_amd.extract_batch(d_imgs[lo:hi], W, d_pts=d_pts[lo * CAP:hi * CAP])
            got.append(c)
        return np.concatenate(got)

    counts = None
    for _ in range(args.warmup):
        counts = step()

    sift_amd.timing_reset()
    sift_amd.timing_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        counts = step()
    barrier()
    elapsed = time.perf_counter() - t0
    sift_amd.timing_enable(False)
    all_counts = None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        k = torch.tensor([float(counts.sum())], dtype=torch.float64)
        dist.all_reduce(k, op=dist.ReduceOp.SUM)
        keypoints_per_step = float(k.item())
        if config4:  # the final host-side gather, in image order
            parts = [None] * world
            dist.all_gather_object(parts, counts.tolist())
            all_counts = np.zeros(args.images, np.int64)
            for r, part in enumerate(parts):
                all_counts[r::world] = part
    else:
        keypoints_per_step = float(counts.sum())
        all_counts = counts if config4 else None

    images_per_step = args.images if config4 else args.batch * world
    value = images_per_step * args.steps / elapsed

    out = {
        "metric": "images/sec at 1080p (keypoints + 128-D descriptors, whole job)",
        "value": value,
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if config4 else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": (f"{args.images} synthetic {W}x{H} {'u8' if args.u8 else 'f32'} images in the whole job (seeds "
                         f"{SEED0}..{SEED0 + args.images - 1}), image i on GPU i mod {world}, batches of {args.batch}, "
                         if config4 else
                         f"{args.batch} synthetic {W}x{H} {'u8' if args.u8 else 'f32'} images per GPU, ")
                        + f"{S} scales, {O} octaves, "
                        f"cap {CAP}, semantics profile {args.profile}{' with doubled first octave' if args.upsample else ''}, "
                        f"inputs resident in HBM"
                        + (", dense blobs" if args.dense else ""),
            "parallelism": f"dp{world} (independent images, no collective; host-side gather of the results)",
        },
        "keypoints_per_s": keypoints_per_step * args.steps / elapsed,
        "keypoints_per_image": keypoints_per_step / images_per_step,
    }
    if config4 and all_counts is not None:
        out["gathered_counts"] = {"images": int(len(all_counts)), "keypoints": int(all_counts.sum()),
                                  "checksum": int((all_counts * (np.arange(len(all_counts)) % 251 + 1)).sum())}

    if rank == 0:
        ab = algorithmic_bytes(S)
        px = sum(octave_pixels(2 * W, 2 * H, O + 1)) if args.upsample else sum(octave_pixels(W, H, O))
        names = {"pyramid": sift_amd.K_PYRAMID, "verify": sift_amd.K_VERIFY,
                 "detect": sift_amd.K_DETECT, "keypoint": sift_amd.K_KEYPOINT}
        kern = {}
        for name, kid in names.items():
            ms, n = sift_amd.timing_read(kid)
            kern[name] = {"ms_total": ms, "launches": n}
        kernel_names = {"pyramid": "pyramid_kernel", "detect": "detect_kernel"}
        # dominant kernel = the one with the most device time
        dom = max(kernel_names, key=lambda k: kern[k]["ms_total"])
        dom_ms, dom_n = kern[dom]["ms_total"], max(kern[dom]["launches"], 1)
        bytes_total = ab[dom] * px * n_local * args.steps  # this rank
        achieved = bytes_total / (dom_ms * 1e-3) if dom_ms > 0 else 0.0
        out["roofline"] = {
            "bound": "hbm",
            "kernel": kernel_names[dom],
            "achieved": achieved / 1e9,
            "peak": HBM_PEAK / 1e9,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK,
            "traffic": None,
            "bytes_per_px": ab[dom],
            "bytes_per_launch": bytes_total / dom_n,
            "avg_launch_ms": dom_ms / dom_n,
            "launches": dom_n,
            "timing": "HIP events around every launch, on the launch stream, inside the timed region",
        }
        if dom == "pyramid":
            out["roofline"]["note"] = ("fused row + column pass: reads the octave's source once and writes the S+3 "
                                       "blur planes, 24 B/px at 2 scales, beside 190 FP32 FMA per pixel")
            # the same kernel against the FP32 vector issue rate (157.3 TFLOP/s data sheet, 2 flops per FMA)
            fma = 38.0 * (S + 3) * px * n_local * args.steps
            out["roofline"]["fp32_issue_frac"] = 2.0 * fma / (dom_ms * 1e-3) / 157.3e12
        out["kernels_ms_per_image"] = {k: v["ms_total"] / (n_local * args.steps) for k, v in kern.items()}
        # HBM-side bytes per launch of the dominant kernel from rocprofv3 counters (separate
        # FETCH_SIZE / WRITE_SIZE passes over this same command, tools/prof_pmc.sh), if committed
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf):
            try:
                t = json.load(open(tf)).get(kernel_names[dom])
                if t and t.get("batch") == args.batch and not config4:
                    if t.get("source_hash") == source_hash():
                        out["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                        out["roofline"]["traffic_source"] = ("rocprofv3 FETCH_SIZE (x2) and WRITE_SIZE passes over this "
                                                             "command on these kernel sources, profiles/pmc_traffic.json")
                    else:  # counted on other kernel sources than the ones in this tree
                        out["roofline"]["traffic_source"] = "stale"
            except (ValueError, OSError):
                pass
        # what this GPU's memory system gives a kernel that only moves bytes, measured in this run
        if not args.no_ceiling:
            ceil_gbps = sift_amd.measure_copy_ceiling(2 << 30, 5)
            out["roofline"]["copy_ceiling_GBps"] = ceil_gbps
            out["roofline"]["frac_of_copy_ceiling"] = achieved / 1e9 / ceil_gbps if ceil_gbps > 0 else None
        if world == 1 and not args.no_photo and not config4 and profile == 0 and S == 2 and not args.upsample:
            out["photo_density"] = photo_density_rate(sift_amd, torch, n=args.batch)
        if world == 1 and args.streaming and not config4:
            out["streaming"] = streaming_rate(sift_amd, torch, d_imgs, args.steps)
        if world == 1 and args.host_fed:
            out["host_fed"] = host_fed_rates(sift_amd, torch, host, d_pts, args.steps)
        if world == 1 and not args.no_cpu_baseline:
            import oracle

            pr = oracle.params(n_octaves=O, n_scales=S, max_points=CAP, profile=profile, upsample=int(args.upsample))
            n_cpu = min(args.cpu_sample, args.batch)
            t0 = time.perf_counter()
            cpu_pts, cpu_counts = oracle.extract_batch(pr, host[:n_cpu], W, n_threads=threads)
            dt = time.perf_counter() - t0
            # the GPU's points of the same images, as sorted bytes (the append order is an atomic's)
            same = bool(np.array_equal(cpu_counts, counts[:n_cpu]))
            for i in range(n_cpu):
                if not same:
                    break
                g = sift_amd.points_to_numpy(d_pts, int(counts[i]), offset=i * CAP).view(oracle.POINT_DTYPE)
                same = oracle.sort_points(g).tobytes() == oracle.sort_points(cpu_pts[i][:cpu_counts[i]]).tobytes()
            out["cpu_baseline"] = {
                "value": n_cpu / dt,
                "unit": "images/s",
                "cores": threads,
                "host_cpus": os.cpu_count(),
                "kind": "port",
                "sample": f"first {n_cpu} images of the same batch, C restatement of the reference "
                          f"kernels, one image per thread",
                "keypoints_match_gpu": same,
                "match": "sorted point records of every sampled image, byte for byte",
            }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
