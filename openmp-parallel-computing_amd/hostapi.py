"""Host-side mirror of the reference's operator interface for the hot path.

Two layers, both thin over the C ABI (abi.py -> libgrayhip.so -> gfx950 kernels):

* the reference's own functions, same names / argument meaning / in-place behaviour, on HOST
  buffers (numpy uint8 arrays standing in for the C `unsigned char *`):
      convert_to_grayscale(data, width, height, channels)      monolithic/include/parallel_to_grayscale.h:4
      sobel_edge(src, dst, width, height)                      monolithic/include/sobel.h:3-5
      gray_passes / pipeline_gray_sobel                        monolithic/src/main.c:33-35, main_with_sobel.c:51-74
* device-resident operators on torch uint8 CUDA tensors (torch is only the allocator / stream
  provider here), used by bench.py and the multi-GPU sharding code.

Nothing here computes pixels in Python; a missing library or GPU raises GrayHipError.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi
from .abi import MODE_FMA, MODE_IEEE, GrayHipError, check  # noqa: F401


def _host_ptr(a: np.ndarray, writable: bool) -> int:
    if not isinstance(a, np.ndarray) or a.dtype != np.uint8 or not a.flags["C_CONTIGUOUS"]:
        raise GrayHipError("host buffers must be C-contiguous numpy uint8 arrays")
    if writable and not a.flags["WRITEABLE"]:
        raise GrayHipError("destination buffer is read-only")
    return a.ctypes.data


# ---- reference-named host functions ---------------------------------------------------------

def convert_to_grayscale(data: np.ndarray, width: int, height: int, channels: int, mode: int | None = None) -> None:
    """In place: R=G=B=(uchar)(0.299f*r+0.587f*g+0.114f*b), alpha untouched."""
    if data.size < width * height * channels:
        raise GrayHipError("buffer smaller than width*height*channels")
    lib = abi.load()
    m = lib.gh_default_mode() if mode is None else mode
    check(lib.gh_convert_to_grayscale_host(_host_ptr(data, True), width, height, channels, m), "convert_to_grayscale")


def sobel_edge(src: np.ndarray, dst: np.ndarray, width: int, height: int) -> None:
    """Writes the interior of dst only (1<=x<w-1, 1<=y<h-1), like the reference."""
    if src.size < width * height or dst.size < width * height:
        raise GrayHipError("buffer smaller than width*height")
    lib = abi.load()
    check(lib.gh_sobel_edge_host(_host_ptr(src, False), _host_ptr(dst, True), width, height), "sobel_edge")


def gray_passes(data: np.ndarray, width: int, height: int, channels: int, passes: int, mode: int = MODE_IEEE) -> float:
    """main.c:33-35 -- `passes` x convert_to_grayscale on one upload; returns kernel seconds."""
    secs = C.c_double(0.0)
    check(abi.load().gh_gray_passes_host(_host_ptr(data, True), width, height, channels, passes, mode, C.byref(secs)),
          "gray_passes")
    return secs.value


def pipeline_gray_sobel(img: np.ndarray, width: int, height: int, channels: int, passes: int = 1,
                        mode: int = MODE_IEEE) -> float:
    """main_with_sobel.c:51-74 -- `passes` x (gray, extract, sobel, replicate) in place; returns kernel seconds."""
    secs = C.c_double(0.0)
    check(abi.load().gh_pipeline_gray_sobel_host(_host_ptr(img, True), width, height, channels, passes, mode, C.byref(secs)),
          "pipeline_gray_sobel")
    return secs.value


def gray_sobel_streamed(rgb: np.ndarray, edge: np.ndarray, tile_rows: int = 4096, mode: int = MODE_IEEE,
                        pinned: bool = False) -> float:
    """Streamed gray+Sobel of a host-resident (h, w, c) frame into a host (h, w) edge plane, band by band with
    H2D / kernel / D2H overlapped on three hipStreams; returns the seconds spent in the pipelined loop."""
    h, w, c = rgb.shape
    if edge.shape != (h, w):
        raise GrayHipError("edge must be (h, w)")
    secs = C.c_double(0.0)
    check(abi.load().gh_gray_sobel_streamed_host(_host_ptr(rgb, False), _host_ptr(edge, True), w, h, c, mode, tile_rows,
                                                 int(pinned), C.byref(secs)), "gray_sobel_streamed")
    return secs.value


def gray_sobel_streamed_rows(frame_rows: np.ndarray, edge: np.ndarray, y0: int, y1: int, tile_rows: int = 4096, mode: int = MODE_IEEE,
                             pinned: bool = False, first_row_of_frame: int = 0, frame_height: int | None = None) -> float:
    """One rank's share of a streamed frame: `frame_rows` is a host (n, w, c) array holding frame rows
    [first_row_of_frame, first_row_of_frame + n), which must include rows y0-1 and y1 unless they lie outside the frame
    (frame_height rows, default: the array is the whole frame).  Rows [y0, y1) are processed into edge ((y1-y0, w) host)."""
    n, w, c = frame_rows.shape
    fh = (first_row_of_frame + n) if frame_height is None else frame_height
    rows = y1 - y0
    if edge.shape != (rows, w):
        raise GrayHipError("edge must be (y1 - y0, w)")
    top, bot = y0 == 0, y1 == fh
    lo, hi = y0 - first_row_of_frame, y1 - first_row_of_frame
    if lo - (0 if top else 1) < 0 or hi + (0 if bot else 1) > n:
        raise GrayHipError("frame_rows does not hold the halo rows of [y0, y1)")
    secs = C.c_double(0.0)
    base = _host_ptr(frame_rows, False) + lo * w * c
    check(abi.load().gh_gray_sobel_streamed_rows_host(base, _host_ptr(edge, True), w, rows, c, mode, tile_rows, int(pinned), int(top), int(bot),
                                                      C.byref(secs)), "gray_sobel_streamed_rows")
    return secs.value


def sum_rgb(rgb: np.ndarray) -> list[int]:
    """Exact per-channel sums of an RGB8 frame (old/parallel_avg_pixel.c restated)."""
    s = (C.c_uint64 * 3)()
    check(abi.load().gh_sum_rgb8_host(_host_ptr(rgb, False), rgb.size // 3, s), "sum_rgb")
    return [int(s[0]), int(s[1]), int(s[2])]


def avg_pixel(rgb: np.ndarray, avgs_in=(0, 0, 0)) -> list[int]:
    """The reference's accumulate-then-divide: avgs = (avgs_in + 3*sum) / (3*N), C integer division."""
    n = rgb.size // 3
    sums = sum_rgb(rgb)
    out = []
    for a, s in zip(avgs_in, sums):
        num = int(a) + 3 * s
        q = abs(num) // (3 * n) if n else 0
        out.append(q if num >= 0 else -q)
    return out


# ---- the reference's historic entry points over `long` samples (old/parallel_*.c) -----------------------------

def _pointer_tree(data: np.ndarray):
    """long*** view of a C-contiguous (rows, cols, 3) int64 array: rows -> cols -> long[3], as the historic callers built it.
    Returns (address of the tree, keep-alive objects)."""
    rows, cols = data.shape[:2]
    base = data.ctypes.data
    cells = (base + (np.arange(rows * cols, dtype=np.uint64) * 24)).astype(np.uint64).reshape(rows, cols)      # &data[r][c][0]
    cells = np.ascontiguousarray(cells)
    row_ptrs = np.ascontiguousarray(cells.ctypes.data + np.arange(rows, dtype=np.uint64) * (8 * cols))         # &cells[r][0]
    return row_ptrs.ctypes.data, (cells, row_ptrs)


def _long_array(a, shape) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.int64)
    if a.shape != shape:
        raise GrayHipError(f"expected an array of shape {shape}, got {a.shape}")
    return a


def _rgb_long_array(a) -> np.ndarray:
    """(rows, cols, 3) longs: the kernels read three longs per pixel and the pointer trees are built with 24-byte cells."""
    a = np.ascontiguousarray(a, dtype=np.int64)
    if a.ndim != 3 or a.shape[2] != 3:
        raise GrayHipError(f"expected a (rows, cols, 3) array, got shape {a.shape}")
    return a


def old_parallel_convolution(img_padded, kernel, convolved_img: np.ndarray, gblur_norm: int) -> np.ndarray:
    """old/parallel_convolution.c:8 -- img_padded (rows+2, cols+2, 3) longs, kernel 3x3 longs; the weighted sums are ADDED to
    convolved_img (rows, cols, 3) int64 and divided by gblur_norm in place."""
    if (not isinstance(convolved_img, np.ndarray) or convolved_img.dtype != np.int64 or not convolved_img.flags["C_CONTIGUOUS"]
            or convolved_img.ndim != 3 or convolved_img.shape[2] != 3):
        raise GrayHipError("convolved_img must be a C-contiguous (rows, cols, 3) int64 array")
    rows, cols = convolved_img.shape[:2]
    img = _long_array(img_padded, (rows + 2, cols + 2, 3))
    k = _long_array(np.asarray(kernel).reshape(9), (9,))
    tree, keep = _pointer_tree(convolved_img)
    check(abi.load().gh_old_parallel_convolution(img.ctypes.data, k.ctypes.data_as(C.POINTER(C.c_long)), tree, rows, cols, int(gblur_norm)),
          "parallel_convolution")
    del keep
    return convolved_img


def old_parallel_avg_pixel(img, avgs_in=(0, 0, 0)) -> list:
    """old/parallel_avg_pixel.c:5 -- avgs = (avgs_in + 3 * sum) / (rows * cols * 3)."""
    img = _rgb_long_array(img)
    rows, cols = img.shape[:2]
    avgs = np.asarray(avgs_in, dtype=np.int64).copy()
    check(abi.load().gh_old_parallel_avg_pixel(img.ctypes.data, avgs.ctypes.data_as(C.POINTER(C.c_long)), rows, cols), "parallel_avg_pixel")
    return [int(v) for v in avgs]


def old_parallel_to_grayscale(img):
    """old/parallel_to_grayscale.c:7 -- returns ((rows, cols, 3) int64 grey image, [min, max])."""
    img = _rgb_long_array(img)
    rows, cols = img.shape[:2]
    out = np.zeros((rows, cols, 3), dtype=np.int64)
    mm = np.zeros(2, dtype=np.int64)
    tree, keep = _pointer_tree(out)
    check(abi.load().gh_old_parallel_to_grayscale(img.ctypes.data, tree, mm.ctypes.data_as(C.POINTER(C.c_long)), rows, cols), "parallel_to_grayscale")
    del keep
    return out, [int(mm[0]), int(mm[1])]


# ---- device-resident operators (torch uint8 CUDA tensors) ------------------------------------

def _dev_ptr(t) -> int:
    import torch
    if not isinstance(t, torch.Tensor) or t.dtype != torch.uint8 or not t.is_cuda or not t.is_contiguous():
        raise GrayHipError("device operands must be contiguous torch.uint8 CUDA tensors")
    return t.data_ptr()


def _stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def init(device: int = 0) -> None:
    check(abi.load().gh_init(device), "gh_init")


def gray_to_plane(rgb, mode: int = MODE_IEEE, out=None):
    """(h, w, c) uint8 -> (h, w) luma plane."""
    import torch
    h, w, c = rgb.shape
    if out is None:
        out = torch.empty((h, w), dtype=torch.uint8, device=rgb.device)
    check(abi.load().gh_gray_rgb8_to_plane_dev(_dev_ptr(rgb), _dev_ptr(out), h * w, c, mode, _stream()), "gray_to_plane")
    return out


def gray_inplace(img, mode: int = MODE_IEEE, passes: int = 1):
    """In-place convert_to_grayscale on a device (h, w, c) frame; passes > 1 = the reference driver's pass loop
    (main.c:33-35) fused into one launch."""
    h, w, c = img.shape
    check(abi.load().gh_gray_inplace_passes_dev(_dev_ptr(img), h * w, c, mode, passes, _stream()), "gray_inplace")
    return img


def sobel(plane, out=None):
    """(h, w) uint8 -> (h, w) Sobel magnitude, border = 0."""
    import torch
    h, w = plane.shape
    if out is None:
        out = torch.empty_like(plane)
    check(abi.load().gh_sobel_u8_dev(_dev_ptr(plane), _dev_ptr(out), w, h, _stream()), "sobel")
    return out


def sobel_of_gray(edge_plane, mode: int = MODE_IEEE, out=None):
    """Pass k >= 2 of the reference's gray + Sobel loop in one launch (gh_sobel_of_gray_u8_dev): (h, w) edge plane of the previous
    pass -> Sobel of lum(e, e, e), border = 0.  Equals sobel(gray_of_plane(edge_plane, mode)) at half the traffic."""
    import torch
    h, w = edge_plane.shape
    if out is None:
        out = torch.empty_like(edge_plane)
    check(abi.load().gh_sobel_of_gray_u8_dev(_dev_ptr(edge_plane), _dev_ptr(out), w, h, mode, _stream()), "sobel_of_gray")
    return out


def sobel_of_gray_rows(buf, out, rows: int, mode: int, top_is_border: bool, bottom_is_border: bool):
    """Band form: `buf` is (rows+2, w) of EDGE rows (rows 1..rows the band, 0 / rows+1 the neighbours' edge rows); `out` is (rows, w)."""
    w = buf.shape[1]
    assert buf.shape[0] == rows + 2 and tuple(out.shape) == (rows, w)
    check(abi.load().gh_sobel_of_gray_u8_rows_dev(_dev_ptr(buf) + w, _dev_ptr(out), w, rows, mode, int(top_is_border),
                                                  int(bottom_is_border), _stream()), "sobel_of_gray_rows")
    return out


def last_kernel() -> str:
    """Which device-kernel form this thread's last Sobel / fused / convolution launch took (gh_last_kernel)."""
    return abi.load().gh_last_kernel().decode()


def sobel_rows(buf, out, rows: int, top_is_border: bool, bottom_is_border: bool):
    """Row band with halo storage: `buf` is (rows+2, w); rows 1..rows are the band, row 0 / rows+1 are
    the halo rows (ignored when the corresponding edge is the frame border).  `out` is (rows, w)."""
    w = buf.shape[1]
    assert buf.shape[0] == rows + 2 and tuple(out.shape) == (rows, w)
    check(abi.load().gh_sobel_u8_rows_dev(_dev_ptr(buf) + w, _dev_ptr(out), w, rows, int(top_is_border),
                                          int(bottom_is_border), _stream()), "sobel_rows")
    return out


def gray_sobel_fused(rgb, mode: int = MODE_IEEE, out=None):
    """(h, w, c) RGB8/RGBA8 -> (h, w) edge plane in one pass over the frame (any width; fastest when width % 16 == 0 and the
    buffers are 16-byte aligned)."""
    import torch
    h, w, c = rgb.shape
    if out is None:
        out = torch.empty((h, w), dtype=torch.uint8, device=rgb.device)
    check(abi.load().gh_gray_sobel_fused_rows_dev(_dev_ptr(rgb), _dev_ptr(out), w, h, c, mode, 1, 1, _stream()),
          "gray_sobel_fused")
    return out


def gray_sobel_fused_rows(buf, out, rows: int, mode: int, top_is_border: bool, bottom_is_border: bool):
    """Band form of the fused kernel: `buf` is (rows+2, w, c) with RGB halo rows 0 and rows+1."""
    w, c = buf.shape[1], buf.shape[2]
    assert buf.shape[0] == rows + 2 and tuple(out.shape) == (rows, w)
    check(abi.load().gh_gray_sobel_fused_rows_dev(_dev_ptr(buf) + w * c, _dev_ptr(out), w, rows, c, mode,
                                                  int(top_is_border), int(bottom_is_border), _stream()),
          "gray_sobel_fused_rows")
    return out


def gray_sobel(rgb, mode: int = MODE_IEEE, out=None):
    """(h, w, c) RGB8/RGBA8 -> (h, w) edge plane, any width and alignment (alias of gray_sobel_fused)."""
    return gray_sobel_fused(rgb, mode, out=out)


def extract_plane(img, out=None):
    import torch
    h, w, c = img.shape
    if out is None:
        out = torch.empty((h, w), dtype=torch.uint8, device=img.device)
    check(abi.load().gh_extract_plane_dev(_dev_ptr(img), _dev_ptr(out), h * w, c, _stream()), "extract_plane")
    return out


def replicate_plane(plane, img):
    h, w, c = img.shape
    check(abi.load().gh_replicate_plane_dev(_dev_ptr(plane), _dev_ptr(img), h * w, c, _stream()), "replicate_plane")
    return img


def gray_of_plane(plane, mode: int = MODE_IEEE, out=None):
    import torch
    if out is None:
        out = torch.empty_like(plane)
    check(abi.load().gh_gray_of_plane_dev(_dev_ptr(plane), _dev_ptr(out), plane.numel(), mode, _stream()), "gray_of_plane")
    return out


def conv3x3(rgb, kernel, norm: int, out=None):
    import torch
    h, w, c = rgb.shape
    if c != 3:
        raise GrayHipError("conv3x3 takes RGB8 frames")
    if out is None:
        out = torch.empty_like(rgb)
    k = (C.c_int32 * 9)(*[int(v) for v in np.asarray(kernel).reshape(-1)])
    check(abi.load().gh_conv3x3_rgb8_dev(_dev_ptr(rgb), _dev_ptr(out), w, h, k, int(norm), _stream()), "conv3x3")
    return out


def conv3x3_rows(buf, out, rows: int, kernel, norm: int, top_is_border: bool, bottom_is_border: bool):
    """Band form of the 3x3 convolution: `buf` is (rows+2, w, 3) with RGB halo rows 0 and rows+1 (ignored where the band edge is the
    frame border: the zero border applies there); `out` is (rows, w, 3)."""
    w, c = buf.shape[1], buf.shape[2]
    if c != 3:
        raise GrayHipError("conv3x3 takes RGB8 frames")
    assert buf.shape[0] == rows + 2 and tuple(out.shape) == (rows, w, 3)
    k = (C.c_int32 * 9)(*[int(v) for v in np.asarray(kernel).reshape(-1)])
    check(abi.load().gh_conv3x3_rgb8_rows_dev(_dev_ptr(buf) + w * 3, _dev_ptr(out), w, rows, k, int(norm), int(top_is_border),
                                              int(bottom_is_border), _stream()), "conv3x3_rows")
    return out


def sum_rgb_dev(rgb):
    """Device-resident exact channel sums -> python ints."""
    import torch
    sums = torch.zeros(3, dtype=torch.int64, device=rgb.device)
    check(abi.load().gh_sum_rgb8_dev(_dev_ptr(rgb), rgb.numel() // 3, sums.data_ptr(), _stream()), "sum_rgb")
    return [int(v) for v in sums.cpu().tolist()]


def mean_gray_minmax_dev(rgb, out=None, minmax=None):
    """g = (r+g+b)/3 replicated; returns (out, minmax) with minmax a device int32[2] = {min, max} (initialised to the reference's
    256 / -1 when not given: old/parallel_to_grayscale.c:9-10) -- the form the sharded reduction consumes without a host round trip."""
    import torch
    if out is None:
        out = torch.empty_like(rgb)
    if minmax is None:
        minmax = torch.tensor([256, -1], dtype=torch.int32, device=rgb.device)
    if rgb.numel():
        check(abi.load().gh_mean_gray_minmax_dev(_dev_ptr(rgb), _dev_ptr(out), rgb.numel() // 3, minmax.data_ptr(), _stream()),
              "mean_gray_minmax")
    return out, minmax


def mean_gray_minmax(rgb, out=None):
    out, mm = mean_gray_minmax_dev(rgb, out)
    lo, hi = mm.cpu().tolist()
    return out, int(lo), int(hi)


def pipeline_gray_sobel_dev(img, passes: int = 1, mode: int = MODE_IEEE):
    """Reference per-pass pipeline on a device-resident (h, w, c) image, in place."""
    import torch
    h, w, c = img.shape
    gray = torch.empty((h, w), dtype=torch.uint8, device=img.device)
    edge = torch.empty((h, w), dtype=torch.uint8, device=img.device)
    check(abi.load().gh_pipeline_gray_sobel_dev(_dev_ptr(img), w, h, c, passes, mode, _dev_ptr(gray), _dev_ptr(edge),
                                                _stream()), "pipeline_gray_sobel_dev")
    return img, gray, edge


def selftest_sobel_mag(variant: int) -> int:
    """Exhaustive device check of one kernel family's magnitude mapping over all (gx, gy); returns #mismatches."""
    bad = C.c_uint64(0)
    check(abi.load().gh_selftest_sobel_mag(variant, C.byref(bad)), "selftest_sobel_mag")
    return int(bad.value)
