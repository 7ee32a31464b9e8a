"""ctypes view of the CPU parity oracle (oracle/sift_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py.  The sift_amd package never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsift_oracle.so")

POINT_DTYPE = np.dtype(
    [("xpos", "<f4"), ("ypos", "<f4"), ("scale", "<f4"), ("orientation", "<f4"), ("data", "<f4", (128,))]
)
assert POINT_DTYPE.itemsize == 528


class Params(C.Structure):
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


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int)
        pp = C.POINTER(Params)
        L.oracle_default_params.argtypes = [pp]
        L.oracle_gaussian1d.argtypes = [fp, C.c_float, C.c_int]
        L.oracle_tables.argtypes = [pp, fp, fp, fp]
        L.oracle_scale_value.argtypes = [pp, C.c_int, C.c_int]
        L.oracle_scale_value.restype = C.c_float
        L.oracle_rotation_table.argtypes = [fp, fp]
        L.oracle_atan2f.argtypes = [C.c_float, C.c_float]
        L.oracle_atan2f.restype = C.c_float
        L.oracle_assign_bin.argtypes = [C.c_float, C.c_int]
        for name in ("oracle_xblur", "oracle_yblur", "oracle_dog", "oracle_hessian", "oracle_gradient"):
            getattr(L, name).argtypes = [pp, fp, fp, C.c_int, C.c_int, C.c_int]
            getattr(L, name).restype = None
        L.oracle_downsample.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int]
        L.oracle_downsample.restype = None
        for name in ("oracle_find_extrema_closed", "oracle_find_extrema_literal"):
            getattr(L, name).argtypes = [pp, fp, fp, C.c_int, C.c_int, C.c_int, ip, C.c_int]
        L.oracle_pixel_rank.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.oracle_orientation.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp]
        L.oracle_orientation.restype = C.c_float
        L.oracle_descriptor.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_float, fp]
        L.oracle_descriptor.restype = None
        L.oracle_extract.argtypes = [pp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, ip]
        L.oracle_extract_batch.argtypes = [pp, fp, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, ip, C.c_int]
        L.oracle_ambiguity_report.argtypes = [pp, fp, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_long)]
        L.oracle_ambiguity_report.restype = None
        L.oracle_gray_q14.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_gray_q14.restype = None
        L.oracle_format_points.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_long]
        L.oracle_format_points.restype = C.c_long
        L.oracle_det_exp2.argtypes = [C.c_double]
        L.oracle_det_exp2.restype = C.c_double
        L.oracle_upsample2x.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp]
        L.oracle_upsample2x.restype = None
        L.oracle_quantize_descriptors.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_quantize_descriptors.restype = None
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def params(n_octaves=4, n_scales=2, max_points=2500, sigma=1.6, r_thresh=10.0, extrema_thresh=10.0,
           profile=0, upsample=0):
    return Params(n_octaves, n_scales, max_points, sigma, r_thresh, extrema_thresh, int(profile), int(upsample))


def gaussian1d(sigma, n):
    out = np.empty(n, np.float32)
    lib().oracle_gaussian1d(_fp(out), sigma, n)
    return out


def tables(pr):
    S = pr.n_scales
    blur = np.empty((S + 3, 19), np.float32)
    wnd = np.empty((S, 16), np.float32)
    desc = np.empty(16, np.float32)
    lib().oracle_tables(C.byref(pr), _fp(blur), _fp(wnd), _fp(desc))
    return blur, wnd, desc


def scale_value(pr, i, o):
    return np.float32(lib().oracle_scale_value(C.byref(pr), i, o))


def rotation_table():
    c = np.empty(32, np.float32)
    s = np.empty(32, np.float32)
    lib().oracle_rotation_table(_fp(c), _fp(s))
    return c, s


def atan2f(y, x):
    return np.float32(lib().oracle_atan2f(float(np.float32(y)), float(np.float32(x))))


def assign_bin(theta, nbins):
    return lib().oracle_assign_bin(float(np.float32(theta)), nbins)


def pitched(img, pitch, fill=np.nan):
    """(h,w) float32 -> (h,pitch) with the pad columns poisoned."""
    h, w = img.shape
    out = np.full((h, pitch), fill, np.float32)
    out[:, :w] = img
    return out


def _planes(n, h, p):
    return np.full((n, h, p), np.nan, np.float32)


def xblur(pr, src, w):
    h, p = src.shape
    tmp = _planes(pr.n_scales + 3, h, p)
    lib().oracle_xblur(C.byref(pr), _fp(np.ascontiguousarray(src)), _fp(tmp), w, p, h)
    return tmp


def yblur(pr, tmp, w):
    n, h, p = tmp.shape
    out = _planes(n, h, p)
    lib().oracle_yblur(C.byref(pr), _fp(np.ascontiguousarray(tmp)), _fp(out), w, p, h)
    return out


def dog(pr, blur, w):
    n, h, p = blur.shape
    out = _planes(pr.n_scales + 2, h, p)
    lib().oracle_dog(C.byref(pr), _fp(np.ascontiguousarray(blur)), _fp(out), w, p, h)
    return out


def hessian(pr, blur, w):
    n, h, p = blur.shape
    out = _planes(3 * pr.n_scales, h, p)
    lib().oracle_hessian(C.byref(pr), _fp(np.ascontiguousarray(blur)), _fp(out), w, p, h)
    return out


def gradient(pr, blur, w):
    n, h, p = blur.shape
    out = _planes(2 * pr.n_scales, h, p)
    lib().oracle_gradient(C.byref(pr), _fp(np.ascontiguousarray(blur)), _fp(out), w, p, h)
    return out


def downsample(plane, w):
    h, p = plane.shape
    w2, h2 = (w + 1) // 2, (h + 1) // 2
    p2 = (w2 + 127) // 128 * 128
    out = np.full((h2, p2), np.nan, np.float32)
    lib().oracle_downsample(_fp(np.ascontiguousarray(plane)), _fp(out), w, p, h)
    return out


def find_extrema(pr, dog3, hess3, w, literal=False, max_out=1 << 20):
    _, h, p = dog3.shape
    xy = np.zeros((max_out, 2), np.int32)
    fn = lib().oracle_find_extrema_literal if literal else lib().oracle_find_extrema_closed
    n = fn(C.byref(pr), _fp(np.ascontiguousarray(dog3)), _fp(np.ascontiguousarray(hess3)), w, p, h,
           xy.ctypes.data_as(C.POINTER(C.c_int)), max_out)
    return xy[: min(n, max_out)].copy()


def orientation(grad2, w, wnd16, x, y):
    _, h, p = grad2.shape
    hist = np.empty(32, np.float32)
    o = lib().oracle_orientation(_fp(np.ascontiguousarray(grad2)), w, p, h,
                                 _fp(np.ascontiguousarray(wnd16)), x, y, _fp(hist))
    return np.float32(o), hist


def descriptor(grad2, w, desc16, x, y, orientation_deg):
    _, h, p = grad2.shape
    out = np.empty(128, np.float32)
    lib().oracle_descriptor(_fp(np.ascontiguousarray(grad2)), w, p, h,
                            _fp(np.ascontiguousarray(desc16)), x, y, float(orientation_deg), _fp(out))
    return out


def extract(pr, img, w=None, octave0=0):
    """img: (h,pitch) float32.  Returns (points[count], meta[count,2], total_found)."""
    img = np.ascontiguousarray(img, np.float32)
    h, p = img.shape
    if w is None:
        w = p
    pts = np.zeros(pr.max_points, POINT_DTYPE)
    meta = np.zeros((pr.max_points, 2), np.uint8)
    total = C.c_int(0)
    n = lib().oracle_extract(C.byref(pr), _fp(img), w, p, h, octave0, pts.ctypes.data,
                             meta.ctypes.data, C.byref(total))
    return pts[:n].copy(), meta[:n].copy(), total.value


def extract_batch(pr, imgs, w=None, n_threads=1):
    """imgs: (n,h,pitch) float32.  Returns (points[n,max_points], counts[n])."""
    imgs = np.ascontiguousarray(imgs, np.float32)
    n, h, p = imgs.shape
    if w is None:
        w = p
    pts = np.zeros((n, pr.max_points), POINT_DTYPE)
    counts = np.zeros(n, np.int32)
    lib().oracle_extract_batch(C.byref(pr), _fp(imgs), h * p, n, w, p, h, pts.ctypes.data,
                               counts.ctypes.data_as(C.POINTER(C.c_int)), n_threads)
    return pts, counts


AMBIGUITY_FIELDS = ("accepted", "accepted_near_edge_limit", "rejected_near_edge_limit",
                    "accepted_near_contrast_limit", "rejected_near_contrast_limit",
                    "edge_decisions_that_depend_on_rounding", "of_those_in_the_output")


def ambiguity_report(pr, img, w=None, tol=1e-3):
    """Threshold decisions of cudaSiftD.cu:288-293 within `tol` of their limit, and edge decisions
    that change under FMA contraction / double-width pow / reciprocal division, as a dict."""
    img = np.ascontiguousarray(img, np.float32)
    h, p = img.shape
    out = (C.c_long * 7)()
    lib().oracle_ambiguity_report(C.byref(pr), _fp(img), p if w is None else w, p, h, tol, out)
    return dict(zip(AMBIGUITY_FIELDS, [int(v) for v in out]))


def gray_q14(rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    out = np.empty(rgb.shape[:-1], np.uint8)
    lib().oracle_gray_q14(rgb.ctypes.data, out.size, out.ctypes.data)
    return out


def format_points(pts):
    pts = np.ascontiguousarray(pts)
    cap = 4096 * (len(pts) + 1)
    buf = C.create_string_buffer(cap)
    n = lib().oracle_format_points(pts.ctypes.data, len(pts), buf, cap)
    assert n >= 0
    return buf.raw[:n].decode()


def quantize_descriptors(pts):
    pts = np.ascontiguousarray(pts)
    out = np.empty((len(pts), 128), np.uint8)
    lib().oracle_quantize_descriptors(pts.ctypes.data, len(pts), out.ctypes.data)
    return out


def upsample2x(img, w=None):
    img = np.ascontiguousarray(img, np.float32)
    h, p = img.shape
    w = p if w is None else w
    out = np.empty((2 * h, 2 * w), np.float32)
    lib().oracle_upsample2x(_fp(img), w, p, h, _fp(out))
    return out


def det_exp2(f):
    return float(lib().oracle_det_exp2(C.c_double(f)))


def sort_points(pts):
    """Canonical order for set comparison: (scale, y, x)."""
    idx = np.lexsort((pts["xpos"], pts["ypos"], pts["scale"]))
    return pts[idx]
