#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in host-pointer entry points (the reference's own signatures) on a 16384^2 RGB8 frame:
never the headline metric (bench.py times device-resident frames), recorded in DESIGN.md section 5.  Developer tool."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("openmp-parallel-computing_amd"); api = pkg.hostapi; api.init(0)
W = H = 16384
for name, make in (("pageable", lambda: np.random.default_rng(1).integers(0, 256, (H, W, 3), dtype=np.uint8)),
                   ("pinned", lambda: torch.randint(0, 256, (H, W, 3), dtype=torch.uint8).pin_memory().numpy())):
    img = make()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); api.convert_to_grayscale(img, W, H, 3); best = min(best, time.perf_counter() - t0)
    print(f"convert_to_grayscale(host ptr, {name:8s}) 16384^2: {best*1e3:8.1f} ms  {W*H/best/1e6:8.0f} Mpixel/s  ({6*W*H/best/1e9:.1f} GB/s over PCIe, both directions)")
    plane = np.ascontiguousarray(img[..., 0]); out = np.zeros_like(plane)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); api.sobel_edge(plane, out, W, H); best = min(best, time.perf_counter() - t0)
    print(f"sobel_edge(host ptrs, {name:8s})          16384^2: {best*1e3:8.1f} ms  {W*H/best/1e6:8.0f} Mpixel/s")
