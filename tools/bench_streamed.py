#!/usr/bin/env python3
"""BASELINE configs[4]: band-streamed gray+Sobel of a host-resident frame (default 65536 x 16384; --height 65536 for the
full 65536^2 = 12.9 GB RGB frame).  Reports Mpixels/s and the PCIe rates; this is PCIe-bound, never the headline metric.
    python tools/bench_streamed.py [--width 65536 --height 16384 --tile-rows 4096 --reps 3]"""
import argparse
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=65536)
    ap.add_argument("--height", type=int, default=16384)
    ap.add_argument("--tile-rows", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--check-rows", type=int, default=64, help="rows spot-checked against the oracle per tile boundary")
    a = ap.parse_args()
    import numpy as np
    import torch
    pkg = importlib.import_module("openmp-parallel-computing_amd")
    api = pkg.hostapi
    api.init(0)
    w, h = a.width, a.height
    t0 = time.time()
    rgb_t = torch.empty((h, w, 3), dtype=torch.uint8).pin_memory()
    edge_t = torch.empty((h, w), dtype=torch.uint8).pin_memory()
    rgb, edge = rgb_t.numpy(), edge_t.numpy()
    # synthetic frame: SplitMix64 stream, generated tile by tile (seed = tile index + 1) like SURVEY.md 8(d)
    from oracle import pyoracle as O
    for k, y0 in enumerate(range(0, h, a.tile_rows)):
        y1 = min(h, y0 + a.tile_rows)
        rgb[y0:y1] = O.frame_rand(w, y1 - y0, 3, seed=k + 1)
    t_gen = time.time() - t0
    best = None
    for _ in range(a.reps):
        secs = api.gray_sobel_streamed(rgb, edge, tile_rows=a.tile_rows, pinned=True)
        best = secs if best is None else min(best, secs)
    # parity: rows around every band boundary (and the frame top) against the oracle
    ok = True
    for y0 in list(range(0, h, a.tile_rows)):
        lo, hi = max(0, y0 - a.check_rows // 2), min(h, y0 + a.check_rows // 2)
        src_lo, src_hi = max(0, lo - 1), min(h, hi + 1)
        want = O.sobel(O.gray_to_plane(np.ascontiguousarray(rgb[src_lo:src_hi]), 0))
        got = edge[lo:hi]
        ok = ok and np.array_equal(got[1:-1], want[lo - src_lo + 1: hi - src_lo - 1])   # rows with both neighbours in the slice
    npix = w * h
    out = {"config": f"{w}x{h} RGB8 streamed in bands of {a.tile_rows} rows, 3 hipStreams, pinned host memory",
           "Mpixels/s": round(npix / best / 1e6, 1), "seconds": round(best, 4),
           "h2d_GB/s": round(3 * npix / best / 1e9, 2), "d2h_GB/s": round(npix / best / 1e9, 2),
           "pcie_total_GB/s": round(4 * npix / best / 1e9, 2), "frame_generation_s": round(t_gen, 1),
           "parity_spot_check": "ok" if ok else "MISMATCH"}
    print(json.dumps(out))
    return 0 if ok else 3


if __name__ == "__main__":
    sys.exit(main())
