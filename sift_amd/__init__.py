"""sift_amd -- MI355X (gfx950) SIFT extractor.

Python view of the C ABI in include/sift_amd.h (libsift_amd.so, built in-tree by
sift_amd/csrc/Makefile).  The function names and argument meaning mirror the
reference host driver (Sift/src/cudaSiftH.h:6-37, Sift/src/sift.h:7-33); torch is
used only to own device memory and streams.  There is no CPU fallback: every
compute call runs the HIP kernels or raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SIFT_AMD_LIB names another build of the same library (kernel experiments); no fallback either way
LIB_PATH = os.environ.get("SIFT_AMD_LIB") or os.path.join(_HERE, "libsift_amd.so")

MAX_SLOTS = 8
K_ROW_BLUR, K_COL_BLUR_DOG, K_DETECT, K_KEYPOINT, K_PYRAMID, K_VERIFY, K_ALIAS = 0, 1, 2, 3, 4, 5, 6

# Sift/src/sift.h:7-13
POINT_DTYPE = np.dtype(
    [("xpos", "<f4"), ("ypos", "<f4"), ("scale", "<f4"), ("orientation", "<f4"), ("data", "<f4", (128,))]
)
POINT_FLOATS = 132


class SiftError(RuntimeError):
    pass


class Params(C.Structure):
    """Runtime mirror of Sift/src/utils.h:4-19."""

    _fields_ = [
        ("n_octaves", C.c_int),
        ("n_scales", C.c_int),
        ("max_points", C.c_int),
        ("sigma", C.c_float),
        ("r_thresh", C.c_float),
        ("extrema_thresh", C.c_float),
        ("profile", C.c_int),
        ("upsample", C.c_int),
    ]


PROFILE_REF, PROFILE_LOWE = 0, 1


# every symbol include/sift_amd.h declares: (restype, argtypes)
_vp, _fp, _ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)
SYMBOLS = {
    "initDeviceConstant": (None, []),
    "initDeviceVariables": (None, []),
    "extractSift": (None, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int]),
    "getPointCount": (C.c_int, [C.c_int]),
    "blurOctave": (None, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "computeDiffOfGauss": (None, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "computeHessian": (None, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "computeGradient": (None, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "copyDeviceData": (None, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "sift_default_params": (None, [C.POINTER(Params)]),
    "sift_set_params": (C.c_int, [C.POINTER(Params)]),
    "sift_get_params": (None, [C.POINTER(Params)]),
    "sift_last_error": (C.c_char_p, []),
    "sift_clear_error": (None, []),
    "sift_set_last_error": (None, [C.c_char_p]),
    "sift_get_tables": (None, [_fp, _fp, _fp]),
    "sift_shutdown": (None, []),
    "sift_release_arenas": (C.c_int, []),
    "sift_extract_batch": (C.c_int, [_vp, _vp, _ip, _vp, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "sift_extract_batch_u8": (C.c_int, [_vp, _vp, _ip, _vp, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "sift_extract_batch_async": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int,
                                           C.c_int, _vp]),
    "sift_extract_noalloc": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "sift_noalloc_scratch_bytes": (C.c_size_t, []),
    "sift_quantize_descriptors": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "sift_set_batch_chunk": (None, [C.c_int]),
    "sift_set_pyramid_columns": (None, [C.c_int, C.c_int]),
    "sift_set_pyramid_split": (None, [C.c_int]),
    "sift_debug_strip_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sift_set_batch_lanes": (None, [C.c_int]),
    "sift_set_queue_cap": (None, [C.c_int]),
    "sift_set_dense_route_pixels": (None, [C.c_long]),
    "sift_get_group_counts": (C.c_int, [C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_int]),
    "sift_debug_read_planes": (C.c_int, [C.c_int, C.c_int, C.c_int, _fp, C.c_long, _ip, _ip, _ip]),
    "sift_debug_batch_slot": (C.c_int, [C.c_int, C.c_int, _ip, _ip]),
    "sift_set_strip_rows": (None, [C.c_int]),
    "sift_device_count": (C.c_int, []),
    "sift_set_device": (C.c_int, [C.c_int]),
    "sift_get_device": (C.c_int, []),
    "sift_dev_malloc": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "sift_dev_malloc_pitch": (C.c_int, [C.POINTER(_vp), _ip, C.c_int, C.c_int]),
    "sift_dev_free": (C.c_int, [_vp]),
    "sift_memcpy2d_h2d_async": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "sift_memcpy2d_d2h_async": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "sift_memcpy_h2d_async": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "sift_memcpy_d2h_async": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "sift_stream_create": (C.c_int, [C.POINTER(_vp)]),
    "sift_stream_destroy": (C.c_int, [_vp]),
    "sift_stream_sync": (C.c_int, [_vp]),
    "sift_host_alloc_pinned": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "sift_host_free_pinned": (C.c_int, [_vp]),
    "sift_convert_u8_f32": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "sift_load_pnm_gray": (C.c_int, [C.c_char_p, C.POINTER(_fp), _ip, _ip]),
    "sift_free_host": (None, [_vp]),
    "sift_format_points": (C.c_long, [_vp, C.c_int, C.c_char_p, C.c_long]),
    "sift_save_points": (C.c_int, [C.c_char_p, C.c_char_p, _vp, C.c_int]),
    "sift": (C.c_int, [C.c_char_p, C.POINTER(C.c_char_p), C.c_int]),
    "sift_sort_points": (None, [_vp, C.c_int]),
    "sift_extract_list_multi": (C.c_int, [C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int, _ip]),
    "sift_list_partition": (C.c_int, [C.c_int, C.c_int]),
    "sift_draw_points": (C.c_int, [C.c_char_p, _fp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _fp]),
    "sift_pipeline_drop_last_point": (None, [C.c_int]),
    "sift_timing_enable": (None, [C.c_int]),
    "sift_timing_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "sift_timing_reset": (None, []),
    "sift_measure_copy_ceiling": (C.c_int, [C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "sift_synth_image": (None, [_vp, C.c_int, C.c_int, C.c_int, C.c_ulonglong, C.c_int]),
}

_lib = None


def lib():
    """Loads libsift_amd.so; raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SiftError(
                f"{LIB_PATH} is missing: build it with `make -C sift_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback"
            )
        # Device memory and streams come from PyTorch, which ships a HIP runtime of its own.  The
        # library must bind to that copy: loaded before torch it binds to /opt/rocm's, the process
        # then holds two runtimes, and the library's sees no usable device.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    return lib().sift_last_error().decode()


def _check():
    msg = last_error()
    if msg:
        lib().sift_clear_error()
        raise SiftError(msg)


def _ok(code):
    if code != 0:
        msg = last_error()
        lib().sift_clear_error()
        raise SiftError(msg or f"error {code}")


# ---- configuration -----------------------------------------------------------
def default_params():
    p = Params()
    lib().sift_default_params(C.byref(p))
    return p


def set_params(n_octaves=4, n_scales=2, max_points=2500, sigma=1.6, r_thresh=10.0, extrema_thresh=10.0,
               profile=PROFILE_REF, upsample=0):
    p = Params(n_octaves, n_scales, max_points, sigma, r_thresh, extrema_thresh, int(profile), int(upsample))
    _ok(lib().sift_set_params(C.byref(p)))
    return p


def get_params():
    p = Params()
    lib().sift_get_params(C.byref(p))
    return p


def get_tables():
    S = get_params().n_scales
    blur = np.empty((S + 3, 19), np.float32)
    wnd = np.empty((S, 16), np.float32)
    desc = np.empty(16, np.float32)
    f = lambda a: a.ctypes.data_as(_fp)
    lib().sift_get_tables(f(blur), f(wnd), f(desc))
    return blur, wnd, desc


def synth_image(width, height, seed, dense=False, pitch=None):
    """Deterministic synthetic image as (height, pitch) float32 (host)."""
    pitch = pitch or width
    img = np.zeros((height, pitch), np.float32)
    lib().sift_synth_image(img.ctypes.data, width, pitch, height, seed, int(dense))
    return img


def load_pnm_gray(path):
    out, w, h = _fp(), C.c_int(), C.c_int()
    rc = lib().sift_load_pnm_gray(os.fsencode(path), C.byref(out), C.byref(w), C.byref(h))
    if rc:
        raise SiftError(f"cannot load {path} (code {rc})")
    try:
        return np.ctypeslib.as_array(out, shape=(h.value, w.value)).copy()
    finally:
        lib().sift_free_host(out)


def extract_list_multi(paths, dst_dir=None, n_workers=1):
    """Every image file of `paths` on n_workers GPUs (image i on worker i mod n_workers); returns the
    keypoint count per image, in list order.  Writes <dst_dir>/<name>.sift.txt when dst_dir is given."""
    arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
    counts = np.zeros(len(paths), np.int32)
    rc = lib().sift_extract_list_multi(os.fsencode(dst_dir) if dst_dir else None, arr, len(paths), n_workers,
                                       counts.ctypes.data_as(_ip))
    if rc:
        msg = last_error() or "sift_extract_list_multi failed"
        lib().sift_clear_error()  # the message travels with the exception, not with the next call
        raise SiftError(msg)
    return counts


def format_points(pts):
    pts = np.ascontiguousarray(pts)
    cap = 2048 * (len(pts) + 1)
    buf = C.create_string_buffer(cap)
    n = lib().sift_format_points(pts.ctypes.data, len(pts), buf, cap)
    if n < 0:
        raise SiftError("format buffer too small")
    return buf.raw[:n].decode()


# ---- device-side entry points (torch tensors own the memory) -----------------
def _stream_ptr(stream=None):
    import torch

    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def _dev_f32(t):
    import torch

    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise SiftError("expected a contiguous float32 CUDA tensor")
    return C.c_void_p(t.data_ptr())


def initDeviceConstant():
    lib().initDeviceConstant()
    _check()


def initDeviceVariables():
    lib().initDeviceVariables()
    _check()


def new_point_buffer(n_images=1, device="cuda"):
    """Device array of n_images * max_points SiftPoint records (as float32 rows of 132)."""
    import torch

    cap = get_params().max_points
    return torch.zeros((n_images * cap, POINT_FLOATS), dtype=torch.float32, device=device)


def extractSift(siftPt, d_src, width, pitch, height, stream=None, streamIdx=0, octaveIdx=0):
    """cudaSiftH.cu:9-60.  siftPt: point buffer tensor; d_src: (height, pitch) float32 CUDA tensor."""
    lib().extractSift(_dev_f32(siftPt), _dev_f32(d_src), width, pitch, height, _stream_ptr(stream),
                      streamIdx, octaveIdx)
    _check()


def group_counts(streamIdx=0):
    """(accepted, queued) per (octave, scale) group of the slot's last extraction."""
    a = (C.c_uint * 32)()
    q = (C.c_uint * 32)()
    n = lib().sift_get_group_counts(streamIdx, a, q, 32)
    if n < 0:
        _check()
        raise SiftError("sift_get_group_counts failed")
    return list(a[:n]), list(q[:n])


def debug_read_planes(slot, image, octave):
    """The S+3 blur planes the pyramid pass left in the arena of `slot` (0..7 = extractSift's
    stream slots, 8 + l = batch lane l) for one image and octave: (planes (S+3, h, pitch), width)."""
    w, p, h = C.c_int(), C.c_int(), C.c_int()
    _ok(lib().sift_debug_read_planes(slot, image, octave, None, 0, C.byref(w), C.byref(p), C.byref(h)))
    n = get_params().n_scales + 3
    out = np.empty((n, h.value, p.value), np.float32)
    _ok(lib().sift_debug_read_planes(slot, image, octave, out.ctypes.data_as(_fp), out.size, C.byref(w),
                                     C.byref(p), C.byref(h)))
    return out, w.value


def debug_batch_slot(n, i):
    """(slot, image) of image i of the latest batch of n images, for debug_read_planes."""
    slot, image = C.c_int(), C.c_int()
    if lib().sift_debug_batch_slot(n, i, C.byref(slot), C.byref(image)):
        raise SiftError(f"image {i} of a batch of {n} is no longer in an arena")
    return slot.value, image.value


def getPointCount(streamIdx=0):
    n = lib().getPointCount(streamIdx)
    if n < 0:
        _check()
        raise SiftError("getPointCount failed")
    return n


def points_to_numpy(siftPt, count, offset=0):
    """First `count` records of a point buffer tensor as a structured numpy array."""
    raw = siftPt[offset:offset + count].cpu().numpy()
    return np.ascontiguousarray(raw).view(POINT_DTYPE).reshape(-1)


def _stage(fn, n_out_planes, src, width, stream):
    import torch

    h, p = src.shape[-2], src.shape[-1]
    dst = torch.full((n_out_planes, h, p), float("nan"), dtype=torch.float32, device=src.device)
    fn(_dev_f32(dst), _dev_f32(src), width, p, h, _stream_ptr(stream))
    _check()
    return dst


def blurOctave(src, width, stream=None):
    """cudaSiftH.cu:210-259: (h,pitch) image -> (S+3, h, pitch) blur planes."""
    return _stage(lib().blurOctave, get_params().n_scales + 3, src, width, stream)


def computeDiffOfGauss(blur, width, stream=None):
    return _stage(lib().computeDiffOfGauss, get_params().n_scales + 2, blur, width, stream)


def computeHessian(blur, width, stream=None):
    return _stage(lib().computeHessian, 3 * get_params().n_scales, blur, width, stream)


def computeGradient(blur, width, stream=None):
    return _stage(lib().computeGradient, 2 * get_params().n_scales, blur, width, stream)


def extract_batch(d_imgs, width, want_meta=False, stream=None, d_pts=None):
    """d_imgs: (n, height, pitch) CUDA tensor, float32 or uint8 (8-bit grey, converted on the
    fly).  Returns (point tensor, counts[, meta])."""
    import torch

    if d_imgs.dim() != 3:
        raise SiftError("expected (n, height, pitch)")
    if d_imgs.dtype not in (torch.float32, torch.uint8) or not d_imgs.is_cuda or not d_imgs.is_contiguous():
        raise SiftError("expected a contiguous float32 or uint8 CUDA tensor")
    n, h, p = d_imgs.shape
    cap = get_params().max_points
    # buffers are created (and the meta bytes zeroed) on the stream the extraction runs on: a fill
    # queued on another stream is not ordered against the kernels that write the buffer
    with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream()):
        if d_pts is None:
            d_pts = torch.empty((n * cap, POINT_FLOATS), dtype=torch.float32, device=d_imgs.device)
        meta = torch.zeros((n * cap, 2), dtype=torch.uint8, device=d_imgs.device) if want_meta else None
    counts = np.zeros(n, np.int32)
    entry = lib().sift_extract_batch_u8 if d_imgs.dtype == torch.uint8 else lib().sift_extract_batch
    rc = entry(_dev_f32(d_pts), C.c_void_p(meta.data_ptr()) if want_meta else None,
               counts.ctypes.data_as(_ip), C.c_void_p(d_imgs.data_ptr()), h * p, n, width, p, h,
               _stream_ptr(stream))
    _ok(rc)
    return (d_pts, counts, meta) if want_meta else (d_pts, counts)


def extract_batch_async(d_imgs, width, d_pts, d_counts, stream=None, d_meta=None):
    """extract_batch without the wait: counts go to d_counts (int32 CUDA tensor of n), nothing is
    read back; ordered on `stream` (None: torch's current stream)."""
    import torch

    if d_imgs.dim() != 3 or d_imgs.dtype not in (torch.float32, torch.uint8) or not d_imgs.is_cuda \
            or not d_imgs.is_contiguous():
        raise SiftError("expected a contiguous (n, height, pitch) float32 or uint8 CUDA tensor")
    n, h, p = d_imgs.shape
    if d_counts.dtype != torch.int32 or d_counts.numel() < n or not d_counts.is_cuda:
        raise SiftError("d_counts: int32 CUDA tensor with one entry per image")
    if d_pts.numel() < n * get_params().max_points * POINT_FLOATS:
        raise SiftError("d_pts is too small")
    _ok(lib().sift_extract_batch_async(_dev_f32(d_pts), C.c_void_p(d_meta.data_ptr()) if d_meta is not None else None,
                                       C.c_void_p(d_counts.data_ptr()), C.c_void_p(d_imgs.data_ptr()),
                                       int(d_imgs.dtype == torch.uint8), h * p, n, width, p, h, _stream_ptr(stream)))


def quantize_descriptors(d_pts, n, stream=None):
    """8-bit descriptors of the first n points of a device point tensor: (n, 128) uint8 tensor."""
    import torch

    out = torch.empty((max(n, 0), 128), dtype=torch.uint8, device=d_pts.device)
    _ok(lib().sift_quantize_descriptors(_dev_f32(d_pts), n, C.c_void_p(out.data_ptr()), _stream_ptr(stream)))
    return out


def timing_enable(on=True):
    lib().sift_timing_enable(int(on))


def timing_reset():
    lib().sift_timing_reset()


def timing_read(kernel):
    ms, n = C.c_double(), C.c_long()
    lib().sift_timing_read(kernel, C.byref(ms), C.byref(n))
    return ms.value, n.value


def measure_copy_ceiling(nbytes=2 << 30, repeats=5):
    """GB/s (read + write) of a plain device copy of nbytes on the current GPU."""
    v = C.c_double()
    _ok(lib().sift_measure_copy_ceiling(nbytes, repeats, C.byref(v)))
    return v.value


def shutdown():
    lib().sift_shutdown()
