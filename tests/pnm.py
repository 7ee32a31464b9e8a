"""Minimal binary PGM/PPM reader for the tests (header comments allowed)."""
import numpy as np


def read_pnm(path):
    b = open(path, "rb").read()
    toks, i = [], 0
    while len(toks) < 4:
        while b[i:i + 1].isspace():
            i += 1
        if b[i:i + 1] == b"#":
            while b[i:i + 1] != b"\n":
                i += 1
            continue
        j = i
        while not b[j:j + 1].isspace():
            j += 1
        toks.append(b[i:j])
        i = j
    i += 1
    w, h = int(toks[1]), int(toks[2])
    ch = 3 if toks[0] == b"P6" else 1
    return np.frombuffer(b[i:i + w * h * ch], np.uint8).reshape(h, w, ch)


def write_pnm(path, arr):
    arr = np.ascontiguousarray(arr, np.uint8)
    if arr.ndim == 2:
        arr = arr[..., None]
    h, w, ch = arr.shape
    with open(path, "wb") as f:
        f.write(b"P6\n" if ch == 3 else b"P5\n")
        f.write(b"# written by the test-suite\n")
        f.write(f"{w} {h}\n255\n".encode())
        f.write(arr.tobytes())
