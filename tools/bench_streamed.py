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
    ap.add_argument("--full-check", type=int, default=1, help="1: EVERY row of this rank's edge plane against the oracle (chunks of 8192 rows, all host cores)")
    a = ap.parse_args()
    import numpy as np
    import torch
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        # N GPUs: the frame's rows are split over the ranks; bands are independent, so the only collective is the
        # timing barrier (gloo).  Launch: python -m torch.distributed.run --nproc-per-node N tools/bench_streamed.py ...
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    pkg = importlib.import_module("openmp-parallel-computing_amd")
    api = pkg.hostapi
    api.init(local % max(torch.cuda.device_count(), 1))
    w, h = a.width, a.height
    y0, y1 = pkg.sharding.row_partition(h, world)[rank]
    lo, hi = max(0, y0 - 1), min(h, y1 + 1)              # this rank's rows plus the halo rows that exist
    t0 = time.time()
    rgb_t = torch.empty((hi - lo, w, 3), dtype=torch.uint8).pin_memory()
    edge_t = torch.empty((y1 - y0, w), dtype=torch.uint8).pin_memory()
    rgb, edge = rgb_t.numpy(), edge_t.numpy()
    # synthetic frame: SplitMix64 stream, generated tile by tile (seed = tile index + 1) like SURVEY.md 8(d); a rank
    # generates the tiles that overlap its rows, so neighbouring ranks agree on their shared halo rows
    from oracle import pyoracle as O
    for k in range(lo // a.tile_rows, (hi - 1) // a.tile_rows + 1):
        t_lo, t_hi = k * a.tile_rows, min(h, (k + 1) * a.tile_rows)
        tile = O.frame_rand(w, t_hi - t_lo, 3, seed=k + 1)
        s_lo, s_hi = max(lo, t_lo), min(hi, t_hi)
        rgb[s_lo - lo:s_hi - lo] = tile[s_lo - t_lo:s_hi - t_lo]
    t_gen = time.time() - t0
    # what the link itself gives: one large pinned host-to-device copy (and, after it, both directions at once), timed with events
    h2d_ceiling = duplex_ceiling = None
    try:
        nb = min(rgb_t.numel(), 2 << 30)
        dev_buf = torch.empty(nb, dtype=torch.uint8, device="cuda")
        back = torch.empty(nb // 3, dtype=torch.uint8).pin_memory()
        src = rgb_t.view(-1)[:nb]
        s2 = torch.cuda.Stream()
        for k in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dev_buf.copy_(src, non_blocking=True)
            if k == 2:
                with torch.cuda.stream(s2):
                    back.copy_(dev_buf[:nb // 3], non_blocking=True)
            e1.record()
            e1.synchronize()
            s2.synchronize()
            gbs = nb / (e0.elapsed_time(e1) * 1e-3) / 1e9
            if k == 1:
                h2d_ceiling = gbs
            if k == 2:
                duplex_ceiling = gbs
        del dev_buf, back
    except Exception:
        pass
    best = None
    for _ in range(a.reps):
        if world > 1:
            dist.barrier()
        t_a = time.perf_counter()
        api.gray_sobel_streamed_rows(rgb, edge, y0, y1, tile_rows=a.tile_rows, pinned=True, first_row_of_frame=lo, frame_height=h)
        secs = time.perf_counter() - t_a
        if world > 1:
            t = torch.tensor([secs], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            secs = float(t.item())
        best = secs if best is None else min(best, secs)
    # parity: rows around every band boundary of this rank (and its first rows) against the oracle
    ok = True
    for b0 in list(range(y0, y1, a.tile_rows)):
        c_lo, c_hi = max(y0, b0 - a.check_rows // 2), min(y1, b0 + a.check_rows // 2)
        src_lo, src_hi = max(lo, c_lo - 1), min(hi, c_hi + 1)
        want = O.sobel(O.gray_to_plane(np.ascontiguousarray(rgb[src_lo - lo:src_hi - lo]), 0))
        got = edge[c_lo - y0:c_hi - y0]
        ok = ok and np.array_equal(got[1:-1], want[c_lo - src_lo + 1: c_hi - src_lo - 1])   # rows with both neighbours in the slice
    # the whole plane of this rank against the CPU restatement (SURVEY.md 8(d) "parity checked per tile ... + sum-of-edge checksum"): the
    # oracle runs on all host cores, a few seconds for 65536^2; the frame has no compiled-reference known answer (per-tile seeds), so the
    # checker is the oracle that the known answers pin at 16384^2
    import hashlib
    full_ok, edge_sum, t_chk = None, int(edge.sum(dtype=np.uint64)), 0.0
    if a.full_check:
        t_c = time.time()
        full_ok = True
        CH = 8192
        for c0 in range(y0, y1, CH):
            c1 = min(y1, c0 + CH)
            s_lo, s_hi = max(lo, c0 - 1), min(hi, c1 + 1)
            want = O.sobel(O.gray_to_plane(np.ascontiguousarray(rgb[s_lo - lo:s_hi - lo]), 0))
            want = want[c0 - s_lo:c1 - s_lo]
            if c0 == 0:
                want[0] = 0                    # the frame's own first / last row: border, written as 0
            if c1 == h:
                want[-1] = 0
            full_ok = full_ok and bool(np.array_equal(edge[c0 - y0:c1 - y0], want))
        t_chk = time.time() - t_c
        ok = ok and full_ok
    digest = hashlib.sha256(edge.data).hexdigest() if world == 1 else None
    if world > 1:
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(t.item())
        t = torch.tensor([edge_sum], dtype=torch.int64)
        dist.all_reduce(t)
        edge_sum = int(t.item())
    npix = w * h
    if rank == 0:
        out = {"config": f"{w}x{h} RGB8 streamed in bands of {a.tile_rows} rows, 3 hipStreams per GPU, pinned host memory, {world} GPU(s), "
                         "rows split over the ranks, no communication",
               "n_gpus": world, "Mpixels/s": round(npix / best / 1e6, 1), "seconds": round(best, 4),
               "h2d_GB/s": round(3 * npix / best / 1e9, 2), "d2h_GB/s": round(npix / best / 1e9, 2),
               "pcie_total_GB/s": round(4 * npix / best / 1e9, 2), "frame_generation_s": round(t_gen, 1),
               "parity_spot_check": "ok" if ok else "MISMATCH",
               "parity_full_frame": (None if full_ok is None else ("ok: every row of the edge plane == oracle" if full_ok else "MISMATCH")),
               "parity_check_s": round(t_chk, 1), "edge_sum": edge_sum, "edge_sha256": digest,
               # this configuration is bound by PCIe H2D (Gen5 x16: 64 GB/s per GPU and direction), not by HBM
               "h2d_copy_ceiling_GB/s": round(h2d_ceiling, 2) if h2d_ceiling else None,
               "h2d_copy_ceiling_with_d2h_beside_GB/s": round(duplex_ceiling, 2) if duplex_ceiling else None,
               "roofline": {"bound": "pcie", "achieved": round(3 * npix / best / 1e9 / world, 2), "peak": 64.0, "unit": "GB/s",
                            "frac": round(3 * npix / best / 1e9 / world / 64.0, 4),
                            "frac_of_copy_ceiling": round(3 * npix / best / 1e9 / world / h2d_ceiling, 4) if h2d_ceiling else None,
                            "what": "host-to-device bytes per GPU (3 B/px) / wall time of the streamed call; the 1 B/px edge plane returns on the other direction concurrently"}}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 3



if __name__ == "__main__":
    sys.exit(main())
