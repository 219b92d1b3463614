"""The aligned-down LDS-DMA rows (UNAL variants of k_sobel_dma / k_fused_dma / k_conv3x3_dma) fetch whole 16-byte blocks around a
strip whose first byte is not 16-byte aligned.  tools/unal_address_model.py restates their address rules (a', which tail blocks)
on the CPU; this test sweeps it: no block may lie entirely outside the readable frame (a GPU page fault otherwise -- a first
version of the convolution's rule did exactly that for rows of 1025 / 1026 bytes), every byte a row step needs must be inside a
fetched block, and the piece must land dword-aligned inside its LDS slot.  The GPU tests (`-m gpu`) check the same kernels'
results for every base offset; this one pins the addressing without a GPU."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_block_outside_the_frame_and_every_needed_byte_fetched():
    spec = importlib.util.spec_from_file_location("unal_address_model", os.path.join(ROOT, "tools", "unal_address_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    n = m.sweep(list(range(1024, 1045)) + [2047, 2048, 2049, 2051, 2064, 3073, 4099, 4100], offsets=range(16))
    assert n > 10000
