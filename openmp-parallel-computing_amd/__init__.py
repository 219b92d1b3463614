"""MI355X-native (gfx950) implementation of the per-pixel hot path of
PedemonteGiacomo/OpenMp-Parallel-Computing: RGB8 -> grayscale, 3x3 Sobel, 3x3 convolution and the
average-pixel reduction, behind the reference's own C signatures.

Layout
    csrc/        hand-written HIP kernels + the C ABI (include/grayhip.h) + the drop-in CLIs
    lib/         libgrayhip.so (built in tree by `make lib`)
    abi.py       ctypes view of the C ABI
    hostapi.py   host-side mirror of the reference interface (convert_to_grayscale, sobel_edge, ...)
    sharding.py  row sharding + one-row halo exchange over torch.distributed (RCCL)

The directory name contains a dash, so import it with importlib:
    pkg = importlib.import_module("openmp-parallel-computing_amd")
"""
from . import abi, hostapi, sharding  # noqa: F401
from .abi import MODE_FMA, MODE_IEEE, GrayHipError  # noqa: F401
from .hostapi import convert_to_grayscale, sobel_edge  # noqa: F401

__all__ = ["abi", "hostapi", "sharding", "convert_to_grayscale", "sobel_edge", "MODE_IEEE", "MODE_FMA", "GrayHipError"]
