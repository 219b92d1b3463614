"""ctypes view of libgrayhip.so -- the C ABI declared in include/grayhip.h.

This is the only way Python reaches the hot path: every call lands in the hand-written gfx950
kernels.  There is no Python/NumPy/torch fallback; if the library is missing or a call fails a
GrayHipError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_PKG_DIR)
LIB_PATH = os.environ.get("GRAYHIP_LIB") or os.path.join(_PKG_DIR, "lib", "libgrayhip.so")   # override: A/B builds
# The measurement lab (`make probes`, -DGH_ENABLE_PROBES): the same library plus the kernel variants that lost their A/B, the
# no-arithmetic probes and the GH_* tuning knobs that select them.  Never the product: tools/ and the variant-sweeping tests load it
# through use_library(); the release library above reads none of those knobs.
LAB_LIB_PATH = os.path.join(REPO_ROOT, "build_probes", "libgrayhip_probes.so")

MODE_IEEE = 0
MODE_FMA = 1

_u8p = C.c_void_p      # device or host byte pointer, passed as an integer address
_stream = C.c_void_p

# name -> (restype, argtypes); kept in step with include/grayhip.h (tests/test_abi_symbols.py checks)
SIGNATURES = {
    "gh_init": (C.c_int, [C.c_int]),
    "gh_device_count": (C.c_int, []),
    "gh_last_error": (C.c_char_p, []),
    "gh_last_kernel": (C.c_char_p, []),
    "gh_version": (C.c_char_p, []),
    "gh_device_topology": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gh_default_mode": (C.c_int, []),
    "gh_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "gh_free": (C.c_int, [C.c_void_p]),
    "gh_memset_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, _stream]),
    "gh_copy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, _stream]),
    "gh_copy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, _stream]),
    "gh_stream_sync": (C.c_int, [_stream]),
    "gh_copy_stream_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, _stream]),
    "gh_gray_rgb8_to_plane_dev": (C.c_int, [_u8p, _u8p, C.c_size_t, C.c_int, C.c_int, _stream]),
    "gh_gray_inplace_dev": (C.c_int, [_u8p, C.c_size_t, C.c_int, C.c_int, _stream]),
    "gh_gray_inplace_passes_dev": (C.c_int, [_u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, _stream]),
    "gh_sobel_u8_dev": (C.c_int, [_u8p, _u8p, C.c_int, C.c_int, _stream]),
    "gh_sobel_u8_rows_dev": (C.c_int, [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, _stream]),
    "gh_sobel_of_gray_u8_dev": (C.c_int, [_u8p, _u8p, C.c_int, C.c_int, C.c_int, _stream]),
    "gh_sobel_of_gray_u8_rows_dev": (C.c_int, [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _stream]),
    "gh_gray_sobel_fused_rows_dev": (C.c_int, [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _stream]),
    "gh_extract_plane_dev": (C.c_int, [_u8p, _u8p, C.c_size_t, C.c_int, _stream]),
    "gh_replicate_plane_dev": (C.c_int, [_u8p, _u8p, C.c_size_t, C.c_int, _stream]),
    "gh_gray_of_plane_dev": (C.c_int, [_u8p, _u8p, C.c_size_t, C.c_int, _stream]),
    "gh_conv3x3_rgb8_dev": (C.c_int, [_u8p, _u8p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int32, _stream]),
    "gh_conv3x3_rgb8_rows_dev": (C.c_int, [_u8p, _u8p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int32, C.c_int, C.c_int, _stream]),
    "gh_sum_rgb8_dev": (C.c_int, [_u8p, C.c_size_t, C.c_void_p, _stream]),
    "gh_mean_gray_minmax_dev": (C.c_int, [_u8p, _u8p, C.c_size_t, C.c_void_p, _stream]),
    "gh_pipeline_gray_sobel_dev": (C.c_int, [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _u8p, _stream]),
    "gh_convert_to_grayscale_host": (C.c_int, [_u8p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gh_gray_passes_host": (C.c_int, [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "gh_sobel_edge_host": (C.c_int, [_u8p, _u8p, C.c_int, C.c_int]),
    "gh_pipeline_gray_sobel_host": (C.c_int, [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "gh_sum_rgb8_host": (C.c_int, [_u8p, C.c_size_t, C.POINTER(C.c_uint64)]),
    "gh_gray_sobel_streamed_host": (C.c_int, [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "gh_comm_unique_id": (C.c_int, [C.c_void_p]),
    "gh_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "gh_comm_rank": (C.c_int, [C.c_void_p]),
    "gh_comm_world": (C.c_int, [C.c_void_p]),
    "gh_comm_destroy": (C.c_int, [C.c_void_p]),
    "gh_comm_set_loopback": (C.c_int, [C.c_void_p, C.c_int]),
    "gh_halo_exchange_dev": (C.c_int, [C.c_void_p, C.c_int, _u8p, _u8p, C.c_int, _u8p, _u8p, C.c_size_t, _stream]),
    "gh_comm_allreduce_sum_u64_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, _stream]),
    "gh_comm_allreduce_minmax_i32_dev": (C.c_int, [C.c_void_p, C.c_void_p, _stream]),
    "gh_comm_exchange_count": (C.c_ulonglong, [C.c_void_p]),
    "gh_comm_halo_timeouts": (C.c_uint, [C.c_void_p]),
    "gh_comm_set_step_form": (C.c_int, [C.c_void_p, C.c_int]),
    "gh_comm_check": (C.c_int, [C.c_void_p]),
    "gh_comm_step_form_in_use": (C.c_int, [C.c_void_p]),
    "gh_comm_side_stream_replacements": (C.c_int, [C.c_void_p]),
    "gh_comm_exchange_alone_us": (C.c_double, [C.c_void_p]),
    "gh_comm_halo_wait_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_double)]),
    "gh_comm_set_halo_wait_ms": (C.c_int, [C.c_void_p, C.c_int]),
    "gh_sharded_conv3x3_step_dev": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int32, _stream]),
    "gh_sharded_pipeline_gray_sobel_dev": (C.c_int, [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _u8p, _stream]),
    "gh_sharded_fused_step_dev": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, _stream]),
    "gh_sharded_sobel_step_dev": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_int, C.c_int, _stream]),
    "gh_sharded_sobel_of_gray_step_dev": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_int, C.c_int, C.c_int, _stream]),
    "gh_sharded_fused_step_pipelined_dev": (C.c_int, [C.c_void_p, _u8p, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, _stream]),
    "gh_sharded_fused_stream_step_dev": (C.c_int, [C.c_void_p, _u8p, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, _stream]),
    "gh_sharded_sobel_stream_step_dev": (C.c_int, [C.c_void_p, _u8p, _u8p, _u8p, C.c_int, C.c_int, _stream]),
    "gh_sharded_sobel_step_pipelined_dev": (C.c_int, [C.c_void_p, _u8p, _u8p, _u8p, C.c_int, C.c_int, _stream]),
    "gh_sharded_prefetch_halos_dev": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_int, _stream]),
    "gh_sharded_prefetch_reset": (C.c_int, [C.c_void_p]),
    "gh_gray_sobel_streamed_rows_host": (C.c_int, [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   C.POINTER(C.c_double)]),
    "gh_old_parallel_convolution": (C.c_int, [C.c_void_p, C.POINTER(C.c_long), C.c_void_p, C.c_int, C.c_int, C.c_long]),
    "gh_old_parallel_avg_pixel": (C.c_int, [C.c_void_p, C.POINTER(C.c_long), C.c_int, C.c_int]),
    "gh_old_parallel_to_grayscale": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_long), C.c_int, C.c_int]),
    "convert_to_grayscale": (None, [_u8p, C.c_int, C.c_int, C.c_int]),
    "sobel_edge": (None, [_u8p, _u8p, C.c_int, C.c_int]),
    "gh_selftest_sobel_mag": (C.c_int, [C.c_int, C.POINTER(C.c_uint64)]),
}


class GrayHipError(RuntimeError):
    pass


_lib = None
_opened = {}          # path -> typed CDLL


def _open(path: str) -> C.CDLL:
    lib = _opened.get(path)
    if lib is None:
        if not os.path.exists(path):
            raise GrayHipError(
                f"{path} is missing: build it with `make lib` / `make probes` (or __graft_entry__.build()); "
                "there is no CPU fallback for the hot path")
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)     # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _opened[path] = lib
    return lib


def load() -> C.CDLL:
    """dlopen libgrayhip.so (no GPU work happens here) and type every exported symbol."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


class use_library:
    """`with abi.use_library(abi.LAB_LIB_PATH):` -- every call of this package goes to another build of the library inside the block
    (tests and tools that sweep kernel variants: the probe build).  Handles made by one library (communicators) must not be handed to
    the other; device memory is the process's and may."""

    def __init__(self, path: str):
        self.path = path

    def __enter__(self):
        global _lib
        self.prev = _lib
        _lib = _open(self.path)
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().gh_last_error().decode(errors="replace")
        raise GrayHipError(f"{what or 'libgrayhip call'} failed (status {rc}): {msg}")
