"""ctypes binding of oracle/liboracle.so (and, when built, oracle/_ref/libref_*.so).

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- as the checker / reported CPU baseline, never as the product
path.  The product path (openmp-parallel-computing_amd/) must not import this module.

All functions take and return numpy uint8 arrays.  The reference lines each function restates
are cited in oracle/pixel_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_DIR = os.path.join(_HERE, "_ref")

MODE_IEEE = 0
MODE_FMA = 1

_u8p = C.POINTER(C.c_uint8)
_lib = None


def build(force: bool = False) -> None:
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.isdir("/root/reference/monolithic/src"):
        subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_luma.restype = C.c_uint8
        L.orc_luma.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8, C.c_int]
        L.orc_gray_inplace.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_gray_to_plane.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_sobel_interior.argtypes = [_u8p, _u8p, C.c_int, C.c_int]
        L.orc_sobel_zero_border.argtypes = [_u8p, _u8p, C.c_int, C.c_int]
        L.orc_extract_plane.argtypes = [_u8p, _u8p, C.c_size_t, C.c_int]
        L.orc_replicate_plane.argtypes = [_u8p, _u8p, C.c_size_t, C.c_int]
        L.orc_pipeline.argtypes = [_u8p, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_gray_passes.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_conv3x3_rgb8.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int32]
        L.orc_sum_rgb8.argtypes = [_u8p, C.c_size_t, C.POINTER(C.c_uint64)]
        L.orc_avg_from_sums.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_int64)]
        L.orc_mean_gray_minmax.argtypes = [_u8p, _u8p, C.c_size_t, C.POINTER(C.c_int32)]
        L.orc_fill_rand.argtypes = [_u8p, C.c_size_t, C.c_uint64]
        L.orc_fill_smooth.argtypes = [_u8p, C.c_int, C.c_int]
        L.orc_fill_cube.argtypes = [_u8p]
        _lp = C.POINTER(C.c_long)
        L.orc_old_parallel_convolution.argtypes = [_lp, _lp, _lp, C.c_int, C.c_int, C.c_long]
        L.orc_old_parallel_avg_pixel.argtypes = [_lp, _lp, C.c_int, C.c_int]
        L.orc_old_parallel_to_grayscale.argtypes = [_lp, _lp, _lp, C.c_int, C.c_int]
        L.orc_isqrt.restype = C.c_uint32
        L.orc_isqrt.argtypes = [C.c_uint32]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u8p)


def sha256(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).data).hexdigest()


# ---- synthetic frames ---------------------------------------------------------------------

def fill_rand(nbytes: int, seed: int = 1) -> np.ndarray:
    buf = np.empty(nbytes, dtype=np.uint8)
    lib().orc_fill_rand(_p(buf), nbytes, seed)
    return buf


def frame_rand(w: int, h: int, channels: int = 3, seed: int = 1) -> np.ndarray:
    return fill_rand(w * h * channels, seed).reshape(h, w, channels)


def frame_smooth(w: int, h: int) -> np.ndarray:
    buf = np.empty((h, w, 3), dtype=np.uint8)
    lib().orc_fill_smooth(_p(buf), w, h)
    return buf


def frame_cube() -> np.ndarray:
    buf = np.empty((4096, 4096, 3), dtype=np.uint8)
    lib().orc_fill_cube(_p(buf))
    return buf


# ---- hot path -------------------------------------------------------------------------------

def gray_inplace(img: np.ndarray, mode: int = MODE_IEEE) -> np.ndarray:
    """a1 on a copy of `img` (h, w, channels)."""
    out = np.ascontiguousarray(img).copy()
    h, w, ch = out.shape
    lib().orc_gray_inplace(_p(out), w, h, ch, mode)
    return out


def gray_to_plane(img: np.ndarray, mode: int = MODE_IEEE) -> np.ndarray:
    img = np.ascontiguousarray(img)
    h, w, ch = img.shape
    out = np.empty((h, w), dtype=np.uint8)
    lib().orc_gray_to_plane(_p(img), _p(out), w, h, ch, mode)
    return out


def sobel(plane: np.ndarray, dst: np.ndarray | None = None) -> np.ndarray:
    """a2.  With dst=None the border is 0 (this build's definition); with a dst array the
    interior is written into it and its border is left untouched, like the reference."""
    plane = np.ascontiguousarray(plane)
    h, w = plane.shape
    if dst is None:
        out = np.empty((h, w), dtype=np.uint8)
        lib().orc_sobel_zero_border(_p(plane), _p(out), w, h)
        return out
    assert dst.shape == plane.shape
    lib().orc_sobel_interior(_p(plane), _p(dst), w, h)
    return dst


def extract_plane(img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img)
    h, w, ch = img.shape
    out = np.empty((h, w), dtype=np.uint8)
    lib().orc_extract_plane(_p(img), _p(out), w * h, ch)
    return out


def replicate_plane(plane: np.ndarray, img: np.ndarray) -> np.ndarray:
    out = np.ascontiguousarray(img).copy()
    h, w, ch = out.shape
    lib().orc_replicate_plane(_p(np.ascontiguousarray(plane)), _p(out), w * h, ch)
    return out


def pipeline(img: np.ndarray, passes: int = 1, mode: int = MODE_IEEE):
    """a5: returns (img_out, gray_plane_of_last_pass, edge_plane_of_last_pass)."""
    out = np.ascontiguousarray(img).copy()
    h, w, ch = out.shape
    gray = np.empty((h, w), dtype=np.uint8)
    edge = np.empty((h, w), dtype=np.uint8)
    lib().orc_pipeline(_p(out), _p(gray), _p(edge), w, h, ch, passes, mode)
    return out, gray, edge


def gray_passes(img: np.ndarray, passes: int, mode: int = MODE_IEEE) -> np.ndarray:
    out = np.ascontiguousarray(img).copy()
    h, w, ch = out.shape
    lib().orc_gray_passes(_p(out), w, h, ch, passes, mode)
    return out


def conv3x3(img: np.ndarray, kernel, norm: int) -> np.ndarray:
    img = np.ascontiguousarray(img)
    h, w, ch = img.shape
    assert ch == 3
    k = (C.c_int32 * 9)(*[int(v) for v in np.asarray(kernel).reshape(-1)])
    out = np.empty_like(img)
    lib().orc_conv3x3_rgb8(_p(img), _p(out), w, h, k, norm)
    return out


def sum_rgb(img: np.ndarray):
    img = np.ascontiguousarray(img)
    n = img.size // 3
    s = (C.c_uint64 * 3)()
    lib().orc_sum_rgb8(_p(img.reshape(-1)), n, s)
    return [int(s[0]), int(s[1]), int(s[2])]


def avg_from_sums(sums, npix: int, avgs_in=(0, 0, 0)):
    s = (C.c_uint64 * 3)(*sums)
    a = (C.c_int64 * 3)(*avgs_in)
    lib().orc_avg_from_sums(s, npix, a)
    return [int(a[0]), int(a[1]), int(a[2])]


def mean_gray_minmax(img: np.ndarray):
    img = np.ascontiguousarray(img)
    n = img.size // 3
    out = np.empty_like(img)
    mm = (C.c_int32 * 2)()
    lib().orc_mean_gray_minmax(_p(img.reshape(-1)), _p(out.reshape(-1)), n, mm)
    return out, int(mm[0]), int(mm[1])


def luma_cube(mode: int = MODE_IEEE) -> np.ndarray:
    """lum for every (r,g,b), indexed [r,g,b]."""
    return gray_to_plane(frame_cube(), mode).reshape(256, 256, 256)


# ---- the compiled reference itself (build container only) ----------------------------------

class RefLib:
    """The reference's own convert_to_grayscale / sobel_edge, compiled in place by oracle/Makefile."""

    def __init__(self, which: str = "ieee"):
        name = {"ieee": "libref_ieee.so", "fma": "libref_fma_native.so"}[which]
        path = os.path.join(REF_DIR, name)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self.L = C.CDLL(path)
        self.L.convert_to_grayscale.argtypes = [_u8p, C.c_int, C.c_int, C.c_int]
        self.L.sobel_edge.argtypes = [_u8p, _u8p, C.c_int, C.c_int]

    def convert_to_grayscale(self, img: np.ndarray) -> np.ndarray:
        out = np.ascontiguousarray(img).copy()
        h, w, ch = out.shape
        self.L.convert_to_grayscale(_p(out), w, h, ch)
        return out

    def sobel_edge(self, plane: np.ndarray) -> np.ndarray:
        plane = np.ascontiguousarray(plane)
        h, w = plane.shape
        out = np.zeros((h, w), dtype=np.uint8)
        self.L.sobel_edge(_p(plane), _p(out), w, h)
        return out


def have_ref(which: str = "ieee") -> bool:
    name = {"ieee": "libref_ieee.so", "fma": "libref_fma_native.so"}[which]
    return os.path.exists(os.path.join(REF_DIR, name))


# ---- the historic functions on `long` arrays (old/parallel_*.c), restated with run-time dimensions ---------------------

def _lp(a: np.ndarray):
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_long))


def old_parallel_convolution(img_padded: np.ndarray, kernel, convolved_inout: np.ndarray, norm: int) -> np.ndarray:
    """img_padded (rows+2, cols+2, 3) int64; convolved_inout (rows, cols, 3) int64, accumulated into and divided in place."""
    rows, cols = convolved_inout.shape[:2]
    assert img_padded.shape == (rows + 2, cols + 2, 3)
    k = np.ascontiguousarray(np.asarray(kernel, dtype=np.int64).reshape(9))
    lib().orc_old_parallel_convolution(_lp(img_padded), _lp(k), _lp(convolved_inout), rows, cols, int(norm))
    return convolved_inout


def old_parallel_avg_pixel(img: np.ndarray, avgs_in=(0, 0, 0)) -> list:
    rows, cols = img.shape[:2]
    avgs = np.asarray(avgs_in, dtype=np.int64).copy()
    lib().orc_old_parallel_avg_pixel(_lp(img), _lp(avgs), rows, cols)
    return [int(v) for v in avgs]


def old_parallel_to_grayscale(img: np.ndarray):
    rows, cols = img.shape[:2]
    out = np.zeros((rows, cols, 3), dtype=np.int64)
    mm = np.zeros(2, dtype=np.int64)
    lib().orc_old_parallel_to_grayscale(_lp(img), _lp(out), _lp(mm), rows, cols)
    return out, [int(mm[0]), int(mm[1])]
