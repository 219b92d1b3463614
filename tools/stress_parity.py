#!/usr/bin/env python3
"""Randomised parity stress of the LDS-DMA kernels and of the sharded step's halo-flag handshake under memory-latency noise (GPU box).

The ring kernels do their own s_waitcnt vmcnt accounting; a wrong count would show up as rare, timing-dependent corrupt
rows.  tests/test_isa_vmcnt.py checks the counts on the generated code; this is the dynamic counterpart: it loops for N
seconds over random shapes (any width) / channel counts / modes / band heights / ring depths / dispatch orders while a second
stream keeps 1 GiB of copy traffic in flight, and compares every result with the oracle.
    usage: stress_parity.py [seconds] [seed]          (a bounded, seeded run is part of `pytest -m gpu`: tests/test_gpu_stress.py)
    GH_STRESS_RELEASE=1: the release library (one kernel form per shape, no knob) instead of the probe build."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOB_SETS = [{}, {"GH_FUSED_DMA": "2"}, {"GH_FUSED_DMA": "3"}, {"GH_FUSED_DMA": "4"}, {"GH_SOBEL_DMA": "2"}, {"GH_SOBEL_DMA": "4"}, {"GH_SOBEL_DMA": "8"},
             {"GH_FUSED_DMA_R": "16", "GH_SOBEL_R": "16"}, {"GH_FUSED_DMA_R": "32", "GH_SOBEL_R": "32"}, {"GH_FUSED_DMA_R": "7", "GH_SOBEL_R": "5"},
             {"GH_FUSED_DMA_ORDER": "0", "GH_SOBEL_ORDER": "0"}, {"GH_FUSED_DMA_ORDER": "4", "GH_SOBEL_ORDER": "4"},
             {"GH_FUSED_DMA_ORDER": "2008", "GH_SOBEL_ORDER": "2006"}, {"GH_FUSED_DMA_PAD_KB": "0"},
             {"GH_FUSED_DMA_UNALIGNED": "0", "GH_SOBEL_DMA_UNALIGNED": "0"}, {"GH_FUSED_DMA_UNALIGNED": "1", "GH_SOBEL_DMA_UNALIGNED": "1"},
             {"GH_SOBEL_NS": "1"}, {"GH_SOBEL_NS": "2"}, {"GH_CONV_R": "16"}, {"GH_CONV_ORDER": "0"}, {"GH_SOBEL_NS": "2", "GH_SOBEL_R": "64"}, {"GH_SOBEL_NS": "2", "GH_SOBEL_R": "16"},
             {"GH_SOBEL_NS": "4"}, {"GH_SOBEL_NS": "4", "GH_SOBEL_NS4_D": "2", "GH_SOBEL_R": "16"}, {"GH_SOBEL_NS": "4", "GH_SOBEL_NS4_CORE": "0", "GH_SOBEL_NS4_D": "6"},
             {"GH_CONV_R": "22"}, {"GH_HALO_MEMOPS": "2"}, {"GH_HALO_MEMOPS": "3"}]


def run(secs: float, seed: int, verbose: bool = True):
    """Returns (results checked, mismatches).  The library re-reads its knobs per launch only under GH_TUNE_DYNAMIC."""
    os.environ["GH_TUNE_DYNAMIC"] = "1"
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch
    from oracle import pyoracle as O          # checker
    pkg = importlib.import_module("openmp-parallel-computing_amd")
    api = pkg.hostapi
    # the knob sets select kernel variants of the probe build (`make probes`); without it the release library runs its one form per shape
    lab = pkg.abi.use_library(pkg.abi.LAB_LIB_PATH) if (os.path.exists(pkg.abi.LAB_LIB_PATH) and not os.environ.get("GH_STRESS_RELEASE")) else None
    if lab is not None:
        lab.__enter__()
    elif verbose:
        print("stress: no probe build (make probes): the release library only", flush=True)
    api.init(0)
    rnd = np.random.RandomState(seed)
    noise_a = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    noise_b = torch.empty_like(noise_a)
    side = torch.cuda.Stream()
    # round 3: the single-frame sharded steps in RCCL loopback (both neighbours are the shard itself) are in the draw -- the halo-flag
    # form's two polled dwords, its upward-walking first band and the wait in front of the halo row meet the same latency noise
    comm = None
    try:
        comm = pkg.sharding.NativeHaloComm(0, 1)
    except Exception as exc:                                  # (no RCCL on this box: the kernels alone)
        if verbose:
            print("stress: no loopback communicator:", exc, flush=True)
    allk = {k for d in KNOB_SETS for k in d} | {"GH_FUSED_UNALIGNED_MIN_PX", "GH_SOBEL_UNALIGNED_MIN_PX"}
    saved = {k: os.environ.get(k) for k in allk}
    t_end, n, bad = time.time() + secs, 0, 0
    t_note = time.time() + 60
    try:
        while time.time() < t_end:
            if verbose and time.time() >= t_note:             # a long run must not look hung
                print(f"... {n} results checked, {bad} mismatches", flush=True)
                t_note = time.time() + 60
            with torch.cuda.stream(side):                       # latency noise: 1 GiB of copy traffic in flight beside the kernels
                for _ in range(4):
                    noise_b.copy_(noise_a, non_blocking=True)
            for k in allk:
                os.environ.pop(k, None)
            os.environ.update(KNOB_SETS[rnd.randint(len(KNOB_SETS))])
            w = 16 * int(rnd.randint(64, 520))
            h = int(rnd.choice([1, 2, 3, 16, 17, 32, 48, 63, 64, 65, 66, 127, 128, 130, 192, 256, 257, 300]))
            ch = int(rnd.choice([3, 3, 4]))
            if rnd.randint(3) == 0:                              # widths that are not multiples of 16
                w += int(rnd.randint(1, 16))
                os.environ["GH_FUSED_UNALIGNED_MIN_PX"] = os.environ["GH_SOBEL_UNALIGNED_MIN_PX"] = "1"
            mode = int(rnd.randint(2))
            img = O.frame_rand(w, h, ch, seed=int(rnd.randint(1 << 30)))
            d = torch.from_numpy(img).cuda()
            luma = O.gray_to_plane(img, mode)
            want = O.sobel(luma)
            outs = [api.gray_sobel_fused(d, mode) for _ in range(3)]
            outs.append(api.sobel(torch.from_numpy(luma).cuda()))
            pass_k = None
            if rnd.randint(3) == 0:                              # pass k of the multi-pass loop: lum(e, e, e) on load, then the window
                pass_k = (api.sobel_of_gray(torch.from_numpy(want).cuda(), mode), O.sobel(O.gray_to_plane(np.repeat(want[:, :, None], 3, axis=2), mode)))
            conv = None
            if ch == 3 and rnd.randint(3) == 0:                  # the 3x3 convolution's ring kernel shares the DMA machinery
                kern, norm = [([1, 2, 1, 2, 4, 2, 1, 2, 1], 16), ([0, -1, 0, -1, 5, -1, 0, -1, 0], 1), ([1] * 9, 9), ([-1, -2, -1, 0, 0, 0, 1, 2, 1], 2)][rnd.randint(4)]
                conv = (api.conv3x3(d, kern, norm), O.conv3x3(img, kern, norm))
            shard = None
            if comm is not None and h >= 3 and rnd.randint(3) == 0:
                # loopback: this rank as a middle (1), top (2) or bottom (3) shard -- a halo row is the shard's own edge row, the other side
                # the frame border; the step in its halo-flag (mostly), serial or event form
                lb = int(rnd.randint(1, 4))
                comm.set_loopback(lb)
                comm.set_step_form(int(rnd.choice([2, 2, 2, 1, 0] if lab is not None else [2, 2, 2, 1])))      # (the event form: probe build only)
                padded = np.concatenate(([luma[:1]] if lb in (1, 3) else []) + [luma] + ([luma[-1:]] if lb in (1, 2) else []), axis=0)
                e_ = O.sobel(padded)
                want_s = e_[(1 if lb in (1, 3) else 0):e_.shape[0] - (1 if lb in (1, 2) else 0)].copy()
                want_s[:, 0] = 0
                want_s[:, -1] = 0
                buf = torch.full((h + 2, w, ch), 0xA5, dtype=torch.uint8, device="cuda")
                buf[1:h + 1] = d
                e1 = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
                lbuf = torch.full((h + 2, w), 0x5A, dtype=torch.uint8, device="cuda")
                lbuf[1:h + 1] = torch.from_numpy(luma).cuda()
                e2 = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
                f1, f2 = comm.bind_fused_step(buf, e1, h, mode), comm.bind_sobel_step(lbuf, e2, h)
                for _ in range(int(rnd.randint(1, 4))):                                # back-to-back steps: step k+1's exchange vs step k's reads
                    f1()
                    f2()
                e3 = None
                if rnd.randint(2) == 0:                                                # the stream step over a ring of two or three copies of the frame
                    ring = [buf] + [buf.clone() for _ in range(int(rnd.randint(1, 3)))]
                    for b_ in ring:
                        b_[0] = 0x3C
                        b_[h + 1] = 0xC3
                    e3 = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
                    f3 = comm.bind_fused_stream_step(ring, e3, h, mode)
                    for k_ in range(int(rnd.randint(1, 6))):
                        f3(k_)
                shard = (e1, e2, want_s) if e3 is None else (e1, e2, want_s, e3)
            torch.cuda.synchronize()
            if shard is not None:
                for e_ in shard[:2] + shard[3:]:
                    n += 1
                    if not np.array_equal(e_.cpu().numpy(), shard[2]):
                        bad += 1
                        if verbose:
                            print("MISMATCH sharded step (loopback mode %d)" % lb, w, h, ch, mode, {k: os.environ.get(k) for k in allk if k in os.environ}, flush=True)
                if comm.halo_timeouts():
                    bad += 1
                    comm.set_step_form(-1)                        # (acknowledges the count: the run goes on)
                    if verbose:
                        print("halo-flag wait timed out", flush=True)
            for o in outs:
                n += 1
                if not np.array_equal(o.cpu().numpy(), want):
                    bad += 1
                    if verbose:
                        print("MISMATCH", w, h, ch, mode, {k: os.environ.get(k) for k in allk if k in os.environ}, flush=True)
            if pass_k is not None:
                n += 1
                if not np.array_equal(pass_k[0].cpu().numpy(), pass_k[1]):
                    bad += 1
                    if verbose:
                        print("MISMATCH sobel_of_gray", w, h, mode, {k: os.environ.get(k) for k in allk if k in os.environ}, flush=True)
            if conv is not None:
                n += 1
                if not np.array_equal(conv[0].cpu().numpy(), conv[1]):
                    bad += 1
                    if verbose:
                        print("MISMATCH conv", w, h, {k: os.environ.get(k) for k in allk if k in os.environ}, flush=True)
    finally:
        if comm is not None:
            comm.close()
        if lab is not None:
            lab.__exit__(None, None, None)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return n, bad


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else (int(time.time()) & 0xffff)
    n, bad = run(secs, seed)
    print(f"{n} results checked in {secs:.0f} s (seed {seed}), {bad} mismatches")
    sys.exit(1 if bad else 0)
