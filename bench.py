#!/usr/bin/env python3
"""bench.py -- the reference's headline workload on MI355X: RGB8 -> grayscale -> 3x3 Sobel of a
synthetic 16384 x 16384 frame (BASELINE.json configs[2]; row-sharded over N GPUs = configs[3]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` (N > 1) without a torch.distributed environment starts its N ranks itself: the parent process, which
never touches the GPU, runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD and relays
rank 0's JSON line and the exit code.

One "step" = one pass of the hot path over one frame that is already resident in HBM:
    --pipeline fused (default): ONE kernel, RGB -> luma -> Sobel edge plane, 3 R + 1 W bytes per pixel
                                (the luma plane never leaves the chip);
    --pipeline split          : gray_rgb8_to_plane (3 R + 1 W B/px) then sobel3x3_u8 (1 R + 1 W B/px).
N = 1: `value` comes from K steps of the chosen pipeline; a second timed region of K steps of the OTHER pipeline follows, and a
third of K steps of the plane Sobel on its own (three rotating luma planes resident in HBM), so that every headline kernel has a
HIP-event figure measured inside a timed region: `kernels.sobel3x3_u8` is the kernel on its own, `kernels.sobel3x3_u8_after_gray`
the same kernel inside the gray -> Sobel pipeline, between two per-step event records (which cost ~2.5 us each: ~4 % on this kernel).
N > 1: the frame is row-sharded, one process per GPU, and every step exchanges a one-row halo with each neighbour over
RCCL: an RGB row for the fused pipeline, a luma row for the split one.  Three step forms are timed, each with the
contract's bracket (barrier + synchronize on both sides, MAX over ranks):
    single_frame  ONE frame, strong-scaled -- BASELINE configs[3] as written, and therefore `value`: per step every rank exchanges
                  its edge rows with its neighbours and filters its shard.  libgrayhip's default form runs the exchange on a side
                  stream beside ONE launch over the shard whose first / last band wait, inside the kernel, for a flag set behind
                  the ncclRecv (csrc/comm.hip; GH_SHARD_STEP_FORM=1: exchange, then launch);
    single_frame_serial_form  the same step with the serial form forced (exchange, then launch): what the halo-flag form has to beat;
    end_to_end    single_frame preceded by the H2D copy of the rank's shard from pinned host memory;
    stream_of_frames  the single-frame step on a ring of eight DIFFERENT frames (no frame seen twice in a row: nothing cached);
    pipelined     a different experiment, reported beside them: the same stream, the halo rows of frame k+1 travelling beside the
                  launch on frame k (gh_sharded_fused_stream_step_dev: the halo-flag handshake, no event between the streams);
    pipelined_batched  round 2's form of it: the halo rows of the next four frames in ONE exchange, events (--halo-batch 0: not timed).
The line also says who issued the exchange (`exchange_fallback` true = torch.distributed P2P instead of libgrayhip's own RCCL
communicator), fails (rc 4) when that communicator did not meet all N ranks, and carries a per-rank breakdown -- shard kernel alone,
exchange alone, the whole step -- so that one run explains the shape of the scaling curve.
Every kernel is the hand-written gfx950 code in libgrayhip.so, called through its C ABI; torch is only
the allocator, the stream provider and the process group.

Prints ONE JSON line on rank 0: Mpixels/s over the whole job, the HBM roofline fraction of the dominant
kernel (HIP-event timed inside the timed region), a per-kernel table, the measured copy ceiling of this GPU and the
reference's OpenMP CPU path timed on this box's host cores (a reported baseline, not the target).
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the achievable rate is measured live (copy_ceiling_GBs)
ALGO_BYTES_PER_PX = {          # each image byte counted once
    "gray_sobel_fused_rgb8": 4,  # 3 R + 1 W
    "gray_rgb8_to_plane": 4,     # 3 R + 1 W
    "sobel3x3_u8": 2,            # 1 R + 1 W
    "sobel3x3_u8_after_gray": 2,  # the same kernel timed inside the gray -> Sobel pipeline (its input plane has just been written)
    "gray_inplace_rgb8": 6,      # 3 R + 3 W (the reference ABI's in-place form)
    "pipeline_pass_k": 2,        # pass k >= 2 of main_with_sobel.c:51-74: the previous edge plane in (lum(e,e,e) applied on load), the next one out
}
M64 = (1 << 64) - 1


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=16384)
    ap.add_argument("--height", type=int, default=16384)
    ap.add_argument("--pipeline", choices=["fused", "split"], default=os.environ.get("GH_BENCH_PIPELINE", "fused"))
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: the 16384^2 frame is row-sharded over N GPUs; weak: every GPU gets `height` rows")
    ap.add_argument("--mode", choices=["ieee", "fma"], default="ieee")
    ap.add_argument("--no-overlap", action="store_true", help="do not overlap the halo exchange with interior rows")
    ap.add_argument("--exchange", choices=["native", "torch"], default=os.environ.get("GH_BENCH_EXCHANGE", "native"),
                    help="N > 1: halo exchange + shard launches issued by libgrayhip's own RCCL communicator from C++ (native) "
                         "or by torch.distributed P2P ops from Python (torch); native falls back to torch if it cannot be set up "
                         "or disagrees with it on the first frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-table", action="store_true", help="skip the second timed region and the per-kernel measurements")
    ap.add_argument("--cpu-budget-s", type=float, default=25.0, help="wall-clock budget of the CPU baseline matrix")
    ap.add_argument("--halo-batch", type=int, default=1, choices=[0, 1],
                    help="N > 1, pipelined form: 1 (default) the halo rows of four future frames per exchange (gh_sharded_prefetch_halos_dev), "
                         "0 one exchange per frame, two frames ahead")
    ap.add_argument("--no-check", action="store_true", help="skip the post-run parity checks")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step bracket of `value` is repeated this many times in all (the first one is `value`); min / median / max in `repeats`")
    ap.add_argument("--edge-ring", type=int, default=3,
                    help="N = 1: the fused (value) region writes its edge plane into this many planes in rotation (default 3: 805 MB of output, beyond the "
                         "256 MB Infinity Cache; 1 = one plane rewritten every step, the A/B)")
    ap.add_argument("--settle-ms", type=float, default=400.0,
                    help="untimed run-in before the W warm-up steps: after ~2 ms of load the GPU's power management dips the clocks and "
                         "takes 30-60 ms to settle (tools/drift_start.py: launches 10-40 of a burst run 25-40 %% slower); "
                         "the timed K steps are meant to see the settled chip")
    return ap.parse_args()


# ---- N > 1 from the plain command: start the ranks as children --------------------------------------------------------

def self_launch(n: int) -> int:
    """`bench.py --gpus N` outside torch.distributed.run: this process has not touched the GPU (torch is not even
    imported) and stays that way; the ranks are children of a child."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "1")                # what torchrun would set anyway (and says so on stderr)
    print(f"bench.py: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode        # stdout is inherited: rank 0 writes the JSON line to it


class Watchdog:
    """N > 1 only.  Every phase that talks to other ranks runs under a wall-clock budget: a peer that never answers leaves a rank inside
    a collective or a device-side receive for good, and a run that is killed from outside leaves no line at all.  When a budget runs out
    rank 0 writes the line with what it has -- `value` and the single-frame figures if they were measured, else `value: null` and the
    reason -- and every rank ends its own process (each has its own watchdog and the same budgets, so they all go within seconds; the
    in-kernel waits of libgrayhip are bounded by themselves, GH_HALO_WAIT_MS).  Exit code 0 when `value` and its checks were good."""

    def __init__(self, json_fd: int, rank: int, minimal: dict):
        import threading
        self.json_fd, self.rank, self.partial, self.good = json_fd, rank, dict(minimal), False
        self.lock = threading.Lock()
        self.deadline, self.phase, self.budget, self.done = None, "", 0.0, False
        self.spent = {}
        self._t0 = None
        self.written = False
        t = threading.Thread(target=self._loop, daemon=True)
        t.start()

    def arm(self, phase: str, seconds: float):
        with self.lock:
            self.phase, self.budget, self.deadline, self._t0 = phase, seconds, time.monotonic() + seconds, time.monotonic()

    def disarm(self):
        with self.lock:
            if self._t0 is not None:
                self.spent[self.phase] = round(self.spent.get(self.phase, 0.0) + time.monotonic() - self._t0, 2)
            self.deadline, self._t0 = None, None

    def finish(self):
        with self.lock:
            self.done, self.deadline = True, None

    def line_written(self):
        """Rank 0 has written the complete line: a later budget (shutdown) ends the process without writing anything."""
        with self.lock:
            self.written = True

    def _loop(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                if self.done:
                    return
                if self.deadline is None or time.monotonic() < self.deadline:
                    continue
                msg = f"{self.phase} exceeded its wall-clock budget of {self.budget:.0f} s on rank {self.rank}"
                print(f"bench.py: watchdog: {msg}; writing what has been measured and ending the process", file=sys.stderr, flush=True)
                if self.rank == 0 and not self.written:
                    line = dict(self.partial)
                    line["watchdog"] = msg
                    line["wall_clock_s"] = dict(self.spent)
                    if line.get("value") is None:
                        line["error"] = msg
                    try:
                        os.write(self.json_fd, (json.dumps(line, default=str) + "\n").encode())
                    except Exception:
                        pass
                os._exit(0 if self.good else 5)


# ---- synthetic frames: SURVEY.md section 8(c) rand(seed), i.e. the SplitMix64 output stream, generated ON the GPU -------

def _s64(v: int) -> int:
    v &= M64
    return v - (1 << 64) if v >> 63 else v


def splitmix_bytes(torch, device, byte0: int, nbytes: int, seed: int):
    """Bytes [byte0, byte0 + nbytes) of rand(seed) as a uint8 tensor (byte0 and nbytes multiples of 8).
    int64 arithmetic wraps mod 2^64 on the device; >> is arithmetic, so logical shifts mask the sign copies off."""
    assert byte0 % 8 == 0 and nbytes % 8 == 0
    out = torch.empty(nbytes, dtype=torch.uint8, device=device)
    words = out.view(torch.int64)
    chunk = 1 << 24
    lsr = lambda x, k: (x >> k) & ((1 << (64 - k)) - 1)
    for w0 in range(0, nbytes // 8, chunk):
        n = min(chunk, nbytes // 8 - w0)
        j = torch.arange(byte0 // 8 + w0 + 1, byte0 // 8 + w0 + 1 + n, dtype=torch.int64, device=device)
        z = j * _s64(0x9E3779B97F4A7C15) + _s64(seed)
        z = (z ^ lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * _s64(0x94D049BB133111EB)
        words[w0:w0 + n] = z ^ lsr(z, 31)
    return out


def known_answers():
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))
    except Exception:
        return {}


def sha256_of(t) -> str:
    h = hashlib.sha256()
    flat = t.reshape(-1)
    step = 1 << 28
    for i in range(0, flat.numel(), step):
        h.update(flat[i:i + step].cpu().numpy().data)
    return h.hexdigest()


# ---- committed PMC evidence ---------------------------------------------------------------------------------------------

def measured_traffic(kernel_name: str, px: int, live_ms: float | None = None, live_form: str = ""):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/<LATEST>/pmc_hbm_traffic.json:
    FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled per the gfx950 correction), scaled to this launch's
    pixel count.  A recorded measurement, not a live one: bench.py cannot run the profiler on itself -- so it is only reported
    while THIS run looks like the run that was profiled: the device kernel the launch rules picked must be the profiled symbol and
    its live, event-timed average duration must lie within 5 % of the profile's average for that symbol (rocprofv3 --stats of the
    same command, kernel_stats.csv beside the counters).  A kernel change that doubled the traffic would change the duration (these
    kernels are HBM-bound) or the symbol; then `traffic` is null and the reason is given instead.  A summary whose two passes
    averaged very different numbers of launches (an outlier filter gone wrong) is refused too.
    Returns (bytes or None, source or reason, cross-check dict)."""
    sym = {"gray_sobel_fused_rgb8": ("k_fused_dma", "k_fused_stream"), "gray_rgb8_to_plane": ("k_gray_rgb_to_plane_w",),
           "sobel3x3_u8": ("k_sobel_dma2", "k_sobel_dma", "k_sobel_stream"), "sobel3x3_u8_after_gray": ("k_sobel_dma2", "k_sobel_dma", "k_sobel_stream")}
    try:
        latest = open(os.path.join(ROOT, "profiles", "LATEST")).read().strip()
        f = os.path.join(ROOT, "profiles", latest, "pmc_hbm_traffic.json")
        rel = os.path.relpath(f, ROOT)
        kernels = json.load(open(f)).get("kernels", {})
        for name in sym[kernel_name]:
            k = kernels.get(name)
            if k and "hbm_total_bytes" in k:
                n = k.get("launches_averaged", [0, 0])
                if min(n) < 1 or max(n) > 10 * min(n):
                    return None, f"{rel} refused: launches averaged {n}", None
                cross = None
                if live_ms is not None:
                    if live_form and not (live_form.split()[0] == name):
                        return None, f"{rel} not applicable: this run launched {live_form!r}, the profile holds {name}", None
                    prof_ns = None
                    try:
                        import csv
                        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", latest, "kernel_stats.csv"))):
                            if f"::{name}<" in r["Name"] or f" {name}<" in r["Name"] or f"::{name}(" in r["Name"]:
                                if prof_ns is None or int(r["Calls"]) > prof_ns[1]:
                                    prof_ns = (float(r["AverageNs"]), int(r["Calls"]))
                    except Exception:
                        prof_ns = None
                    if prof_ns is None:
                        return None, f"{rel} not cross-checked: no kernel_stats.csv entry for {name}", None
                    scale = px / (16384 * 16384)
                    ratio = live_ms * 1e6 / (prof_ns[0] * scale)
                    cross = {"live_avg_us": round(live_ms * 1e3, 2), "profiled_avg_us": round(prof_ns[0] * scale / 1e3, 2), "ratio": round(ratio, 4),
                             "tolerance": 0.05, "profile": f"profiles/{latest}/kernel_stats.csv"}
                    if abs(ratio - 1.0) > 0.05:
                        return None, f"{rel} withheld: live launch duration is {ratio:.3f} x the profiled one (tolerance 5 %)", cross
                return int(k["hbm_total_bytes"] / (16384 * 16384) * px), rel, cross
    except Exception:
        pass
    return None, None, None


# ---- the reference's CPU path on this box's host cores ------------------------------------------------------------------

def host_topology() -> dict:
    """Logical CPUs this process may run on, the physical cores among them (one logical CPU per (package, core id), as
    `lscpu` counts them -- bench_and_plot_monolithic.sh:12), the cgroup CPU quota if there is one, and the CPU model."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cores = {}
    for c in allowed:
        try:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            key = (open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip())
        except OSError:
            key = ("?", str(c))
        cores.setdefault(key, c)
    quota = None
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(f).read().split()
            if f.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"logical": len(allowed), "physical": len(cores), "cgroup_cpus": quota, "model": model,
            "first_cpu_of_each_core": sorted(cores.values())}


def cpu_has(*flags) -> bool:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                have = set(line.split(":", 1)[1].split())
                return all(f in have for f in flags)
    except OSError:
        pass
    return False


def cpu_baseline(width: int, height: int, budget_s: float, topo: dict | None = None) -> dict:
    """SURVEY.md section 8(d) matrix: {1 thread, all physical cores} x {service flags `-O3 -fopenmp`
    (microservices/grayscale/c/Makefile:2), monolithic flags `-O3 -ffast-math -funroll-loops -fopenmp`
    (monolithic/Makefile:2; built for x86-64-v3 instead of -march=native so that the file runs on another host)},
    kernel-only timing around the calls like monolithic/src/main.c:30-38, best of 5 with the spread.
    The code timed is the reference's own parallel_to_grayscale.c + sobel.c compiled where they lie (oracle/_ref/*.so,
    kind "reference") when those prebuilt files travelled with the repo, otherwise this repo's C restatement (kind "port").
    OMP_PLACES=cores / OMP_PROC_BIND=close were exported at the top of main(), before any OpenMP runtime was loaded; the frame
    is first-touched by the same static partition of threads that later processes it.  The all-cores cells run on the whole
    frame, the 1-thread cells on a row sample so that the matrix stays inside its time budget."""
    import ctypes as C

    import numpy as np
    from oracle import pyoracle as O

    t_end = time.time() + budget_s
    topo = topo or host_topology()
    nthreads_all = topo["physical"]
    if topo["cgroup_cpus"]:
        nthreads_all = max(1, min(nthreads_all, int(topo["cgroup_cpus"])))
    gomp = C.CDLL("libgomp.so.1")
    L = O.lib()
    u8p = C.POINTER(C.c_uint8)
    libs = {}
    if O.have_ref("ieee"):
        libs["service flags (-O3 -fopenmp)"] = ("reference", O.RefLib("ieee").L)
    fast = os.path.join(O.REF_DIR, "libref_fast_v3.so")
    if os.path.exists(fast) and cpu_has("avx2", "fma", "bmi2"):
        lib = C.CDLL(fast)
        lib.convert_to_grayscale.argtypes = [u8p, C.c_int, C.c_int, C.c_int]
        lib.sobel_edge.argtypes = [u8p, u8p, C.c_int, C.c_int]
        libs["monolithic flags (-O3 -march=x86-64-v3 -ffast-math -funroll-loops -fopenmp)"] = ("reference", lib)
    if not libs:
        class Port:      # this repo's restatement behind the reference's two prototypes
            @staticmethod
            def convert_to_grayscale(p, w, h, c):
                L.orc_gray_inplace(p, w, h, c, O.MODE_IEEE)

            @staticmethod
            def sobel_edge(s, d, w, h):
                L.orc_sobel_interior(s, d, w, h)
        libs["port (-O3 -fopenmp -ffp-contract=off)"] = ("port", Port)

    def make_frame(rows, threads):
        gomp.omp_set_num_threads(threads)
        n = width * rows * 3
        img = np.empty(n, dtype=np.uint8)                   # untouched pages ...
        L.orc_fill_rand(img.ctypes.data_as(u8p), n, 1)      # ... first-touched by the static partition of `threads` threads
        plane = np.empty(width * rows, dtype=np.uint8)
        L.orc_gray_to_plane(img.ctypes.data_as(u8p), plane.ctypes.data_as(u8p), width, rows, 3, O.MODE_IEEE)
        edge = np.empty(width * rows, dtype=np.uint8)
        L.orc_gray_to_plane(img.ctypes.data_as(u8p), edge.ctypes.data_as(u8p), width, rows, 3, O.MODE_IEEE)   # first touch of dst
        return img, plane, edge

    def time_cell(lib, rows, threads, frame, deadline):
        img, plane, edge = frame
        gomp.omp_set_num_threads(threads)
        tg, ts = [], []
        for rep in range(6):
            t0 = time.perf_counter()
            lib.convert_to_grayscale(img.ctypes.data_as(u8p), width, rows, 3)      # in place; later reps see grey input: same work
            t1 = time.perf_counter()
            lib.sobel_edge(plane.ctypes.data_as(u8p), edge.ctypes.data_as(u8p), width, rows)
            t2 = time.perf_counter()
            if rep:                                                                # first repetition = warm-up
                tg.append(t1 - t0)
                ts.append(t2 - t1)
            if time.time() > deadline and len(tg) >= 2:
                break
        npx = width * rows
        best = min(g + s for g, s in zip(tg, ts))
        worst = max(g + s for g, s in zip(tg, ts))
        return {"threads": threads, "rows": rows, "Mpixels/s": round(npx / best / 1e6, 1), "worst_Mpixels/s": round(npx / worst / 1e6, 1),
                "gray_Mpixels/s": round(npx / min(tg) / 1e6, 1), "sobel_Mpixels/s": round(npx / min(ts) / 1e6, 1), "reps": len(tg)}

    rows1 = min(height, 2048)                     # 1-thread sample: 33.5 Mpx, ~0.25 s per repetition
    matrix, kind = {}, "port"
    share = max((t_end - time.time()) / (2 * len(libs)), 1.0)
    frame_all = make_frame(height, nthreads_all)
    for name, (k, lib) in libs.items():
        kind = k
        matrix[f"{name}, {nthreads_all} threads"] = time_cell(lib, height, nthreads_all, frame_all, time.time() + share)
    del frame_all
    if nthreads_all > 1:
        frame_one = make_frame(rows1, 1)
        for name, (k, lib) in libs.items():
            matrix[f"{name}, 1 thread"] = time_cell(lib, rows1, 1, frame_one, time.time() + share)
    head_key = next(iter(matrix))                  # service flags on all physical cores: what images/test_gray.png was made with
    head = matrix[head_key]
    return {
        "value": head["Mpixels/s"],
        "unit": "Mpixels/s",
        "cores": nthreads_all,
        "kind": kind,
        "sample": f"the whole {width}x{height} rand(seed=1) frame ({width * height / 1e6:.0f} Mpx) for the all-cores cells, its first {rows1} rows "
                  f"for the 1-thread cells; convert_to_grayscale + sobel_edge, kernel-only, best of 5 (worst also given); "
                  f"value = {head_key}",
        "matrix": matrix,
        "host": {k: topo[k] for k in ("logical", "physical", "cgroup_cpus", "model")},
        "omp": {"OMP_PLACES": os.environ.get("OMP_PLACES"), "OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND")},
    }


def main() -> int:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)
    # stdout carries exactly ONE line, the JSON: libraries print banners to file descriptor 1 (RCCL's version block when a
    # communicator is created, gloo's connection notes), so everything else is sent to stderr from here on
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    topo = None
    if world == 1:
        # the CPU baseline's thread placement must be in the environment before ANY OpenMP runtime is loaded (torch brings one);
        # the host topology is read first: once OMP_PROC_BIND is in force the main thread is bound to ONE core and
        # sched_getaffinity() would report just that core
        topo = host_topology()
        os.environ.setdefault("OMP_PLACES", "cores")
        os.environ.setdefault("OMP_PROC_BIND", "close")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the hot path has no CPU fallback", file=sys.stderr)
        return 1
    # GH_BENCH_REHEARSAL=1: several ranks share the GPUs that are present (a one-GPU box), process group over gloo.
    # Only for exercising the multi-rank code path; RCCL refuses two ranks on one device, so the exchange falls back
    # to torch.distributed P2P (through host memory) and the numbers mean nothing.
    # GH_BENCH_REHEARSAL=loopback: as above, and every rank drives the NATIVE step code path with a private one-rank
    # communicator in loopback mode (its neighbours are itself), results unchecked.
    rehearsal = os.environ.get("GH_BENCH_REHEARSAL") in ("1", "loopback")
    rehearsal_loopback = os.environ.get("GH_BENCH_REHEARSAL") == "loopback"
    if rehearsal:
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    METRIC = "Mpixels/s, 16384x16384 RGB8 grayscale+Sobel"
    wd = None
    budget_scale = float(os.environ.get("GH_BENCH_BUDGET_SCALE", "1"))
    if world > 1:
        wd = Watchdog(json_fd, rank, {"metric": METRIC, "value": None, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                                      "ms_per_step": None, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u8",
                                      "data": "synthetic"})

    class phase:
        """`with phase(name, seconds):` -- the block runs under the watchdog's wall-clock budget (N > 1; a no-op on one GPU)"""

        def __init__(self, name, seconds):
            self.name, self.seconds = name, seconds * budget_scale

        def __enter__(self):
            if wd is not None:
                wd.arm(self.name, self.seconds)

        def __exit__(self, *exc):
            if wd is not None:
                wd.disarm()
            return False

    pkg = importlib.import_module("openmp-parallel-computing_amd")
    api, sharding = pkg.hostapi, pkg.sharding
    api.init(local_rank)
    lib = pkg.abi.load()
    import ctypes as _C
    _cu, _xcd = _C.c_int(0), _C.c_int(0)
    pkg.abi.check(lib.gh_device_topology(_C.byref(_cu), _C.byref(_xcd)), "gh_device_topology")
    device_topology = {"compute_units": _cu.value, "xcds": _xcd.value}       # what the launch rules size their plans for
    mode = pkg.MODE_FMA if args.mode == "fma" else pkg.MODE_IEEE

    W = args.width
    H_total = args.height * (world if args.scaling == "weak" else 1)
    y0, y1 = sharding.row_partition(H_total, world)[rank]
    rows = y1 - y0
    geom = sharding.ShardGeometry(rank, world, W, y0, y1)
    fused = args.pipeline == "fused"
    overlap = not args.no_overlap
    backend = sharding.HipBackend(mode)
    stream = torch.cuda.current_stream()
    single = world == 1
    px_rank = rows * W

    def all_ranks_agree(flag: bool) -> bool:
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step_fn, k_steps, events=None, region_events=None):
        """The contract's bracket: barrier + synchronize, EXACTLY k_steps steps, barrier + synchronize; MAX over ranks.
        Returns (seconds, host seconds spent enqueueing).  `events`: per-step HIP events the step function records (needed to tell
        two kernels of one step apart; every record costs ~2.5 us of GPU time between launches); `region_events`: ONE pair recorded
        on the launch stream around the K steps -- the kernel's average launch duration without any instrumentation between launches."""
        fence()
        t0 = time.perf_counter()
        if region_events:
            region_events[0].record(stream)
        for k in range(k_steps):
            step_fn(k, events[k] if events else None)
        if region_events:
            region_events[1].record(stream)
        host = time.perf_counter() - t0
        fence()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()), host

    def run_in(step_fn):
        if args.settle_ms > 0:       # run-in (untimed, before the warm-up steps): let the clocks settle under this very load
            t_end = time.perf_counter() + args.settle_ms * 1e-3
            k = 0
            while True:
                for _ in range(16):
                    step_fn(k, None)
                    k += 1
                torch.cuda.synchronize()
                done = time.perf_counter() >= t_end
                if world > 1:        # every step is an exchange with the neighbours: all ranks must run the SAME number of steps
                    done = all_ranks_agree(done)
                if done:
                    break
        for k in range(args.warmup):
            step_fn(k, None)

    # ranks that really met: one all-reduce of a 1 over the process group, and (native path) one over libgrayhip's own RCCL communicator
    ranks_met = 1
    if world > 1:
        t = torch.ones(1, dtype=torch.int64, device=device)
        dist.all_reduce(t)
        ranks_met = int(t.item())

    # N > 1: who issues the halo exchange.  native = libgrayhip's own RCCL communicator, whole step in one C++ call.
    native, exchange, exchange_note = None, "none", ""
    step_form = None
    comms_to_close = []
    rccl_ranks = None
    if world > 1:
        exchange = "torch.distributed P2P (batch_isend_irecv over RCCL)"
        if args.exchange == "native":
            with phase("libgrayhip communicator setup (ncclCommInitRank + neighbour warm-up + one all-reduce)", 240):
                try:
                    if rehearsal_loopback:
                        # every rank drives the native step with a private one-rank communicator, as the kind of shard it is in the job:
                        # rank 0 the top shard (frame border above, one halo row), the last rank the bottom shard, the others middle shards
                        native = sharding.NativeHaloComm(0, 1)
                        native.set_loopback(native.LOOPBACK_TOP if rank == 0 else (native.LOOPBACK_BOTTOM if rank == world - 1 else native.LOOPBACK_MIDDLE))
                    else:
                        native = sharding.NativeHaloComm(rank, world)
                except Exception as exc:
                    exchange_note = f"native communicator unavailable on rank {rank}: {exc}"
                if native is not None:
                    comms_to_close.append(native)
                if not all_ranks_agree(native is not None):
                    native = None
                if native is not None:
                    one = torch.ones(1, dtype=torch.int64, device=device)
                    native.allreduce_sum_u64(one, stream.cuda_stream)
                    torch.cuda.synchronize()
                    rccl_ranks = int(one.item())
            # the line must never quietly time torch P2P when libgrayhip's communicator was asked for: on real ranks that is a failure
            if not rehearsal and (native is None or rccl_ranks != world):
                msg = f"libgrayhip's RCCL communicator met {rccl_ranks} of {world} ranks" if native is not None else \
                    f"libgrayhip's RCCL communicator could not be created ({exchange_note or 'on another rank'})"
                print(f"bench.py: {msg}; refusing to time the torch.distributed fallback in its place (--exchange torch asks for it explicitly)",
                      file=sys.stderr)
                if rank == 0:
                    os.write(json_fd, (json.dumps({"metric": METRIC, "value": None, "unit": "Mpixels/s",
                                                   "n_gpus": world, "error": msg, "ranks": ranks_met, "rccl_ranks": rccl_ranks,
                                                   "exchange_fallback": True}) + "\n").encode())
                if wd is not None:
                    wd.finish()
                dist.barrier()
                for comm in comms_to_close:
                    try:
                        comm.close()
                    except Exception:
                        pass
                dist.destroy_process_group()
                return 4

    # Synthetic frames, resident in HBM before the timed region: SURVEY.md 8(c) rand(seed), this rank's rows of it.
    # N = 1: one frame, rand(seed=1).  N > 1: a ring of eight frames, rand(seed = 1 .. 8) (a stream of frames, step k works
    # on frame k % 8), so that the halo rows of later frames can be exchanged while the current one is processed -- four frames
    # per exchange in the batched form -- and so that the ring (8 x 100 MB per rank at N = 8) cannot live in the 256 MB Infinity Cache.
    if wd is not None:
        wd.arm("synthetic frames and shard runners", 240 * budget_scale)
    NBUF = 1 if single else 8
    BATCH = 4
    row_bytes = W * 3
    aligned8 = (y0 * row_bytes) % 8 == 0 and (rows * row_bytes) % 8 == 0
    runners, rgbs = [], []
    for i in range(NBUF):
        if aligned8:
            frame = splitmix_bytes(torch, device, y0 * row_bytes, rows * row_bytes, seed=1 + i).view(rows, W, 3)
        else:                        # odd geometry given on the command line: any bytes will do for timing
            gen = torch.Generator(device=device)
            gen.manual_seed(1234 + rank + 1000 * i)
            frame = torch.randint(0, 256, (rows, W, 3), dtype=torch.uint8, device=device, generator=gen)
        if single:
            rgbs.append(frame)
            continue
        if fused:
            r = sharding.ShardedFusedGraySobel(geom, backend, device, channels=3, overlap=overlap)
            r.shard.copy_(frame)
            rgbs.append(r.shard)
            del frame
        else:
            r = sharding.ShardedGraySobel(geom, backend, device, overlap=overlap)
            rgbs.append(frame)
        runners.append(r)
    if wd is not None:
        wd.disarm()
    rgb = rgbs[0]
    halo_bufs = ([r.rgb for r in runners] if fused else [r.luma for r in runners]) if not single else []   # (rows+2)-row buffers
    KA = known_answers()
    is_known_frame = aligned8 and W == 16384 and H_total == 16384
    ka_mode = "FMA" if mode == pkg.MODE_FMA else "IEEE"

    out = {}
    check = "skipped"
    halo_ok = True
    halo_rows_note = None

    if single:
        # ================================ one GPU =========================================================================
        luma = torch.zeros((rows + 2, W), dtype=torch.uint8, device=device)
        # The value region's output rotates over THREE edge planes (805 MB in all): one 268 MB plane rewritten step after step would fit the
        # 256 MB Infinity Cache almost whole, and FETCH_SIZE / WRITE_SIZE count fabric requests including its hits (MI355X_MICROARCH.md) --
        # `--edge-ring 1` is the A/B against that (profiles/r4_ab/fused_edge_ring.txt: the same within noise, the stores being nontemporal).
        edge_ring = [torch.empty((rows, W), dtype=torch.uint8, device=device) for _ in range(max(args.edge_ring, 1))]
        edge = edge_ring[0]
        rgb_ptr, edge_ptr, luma_ptr = rgb.data_ptr(), edge.data_ptr(), luma.data_ptr()
        edge_ring_ptr, n_ring = [t.data_ptr() for t in edge_ring], len(edge_ring)
        s_ptr = stream.cuda_stream
        chk = pkg.abi.check

        def step_fused(k, ev):
            if ev:
                ev[0].record(stream)
            chk(lib.gh_gray_sobel_fused_rows_dev(rgb_ptr, edge_ring_ptr[k % n_ring], W, rows, 3, mode, 1, 1, s_ptr), "fused")
            if ev:
                ev[-1].record(stream)

        def step_split(k, ev):
            if ev:
                ev[0].record(stream)
            chk(lib.gh_gray_rgb8_to_plane_dev(rgb_ptr, luma_ptr + W, rows * W, 3, mode, s_ptr), "gray")
            if ev:
                ev[1].record(stream)
            chk(lib.gh_sobel_u8_rows_dev(luma_ptr + W, edge_ptr, W, rows, 1, 1, s_ptr), "sobel")
            if ev:
                ev[-1].record(stream)

        first, second = (step_fused, step_split) if fused else (step_split, step_fused)
        n_first, n_second = (2, 3) if fused else (3, 2)
        # One kernel per step (fused): a single event pair around the K steps -- nothing between the launches, the average launch
        # duration includes the ~1.5 us between two dependent launches and nothing else.  Two kernels per step (split): per-step
        # events, which cost the step ~5 us (round 2 until here timed the fused region that way too: 0.1813 ms by events, 0.1868 by
        # the wall clock -- `value` paid for the instrumentation).
        def region_pair():
            return [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]

        ev1 = None if fused else [[torch.cuda.Event(enable_timing=True) for _ in range(n_first)] for _ in range(args.steps)]
        rv1 = region_pair()
        run_in(first)
        elapsed, host_enqueue = timed(first, args.steps, ev1, rv1)
        step_ms_events = rv1[0].elapsed_time(rv1[1]) / args.steps
        # the same bracket again (same K, no run-in between): box noise made visible -- `value` stays the FIRST bracket
        rep_ms = [elapsed / args.steps * 1e3]
        for _ in range(max(args.repeats, 1) - 1):
            el_r, _ = timed(first, args.steps, None, None)
            rep_ms.append(el_r / args.steps * 1e3)
        timed_kernels, region_of = {}, {}

        def note(events, is_fused, region, pair=None):
            if is_fused:
                timed_kernels["gray_sobel_fused_rgb8"] = pair[0].elapsed_time(pair[1]) / args.steps
                region_of["gray_sobel_fused_rgb8"] = region
            else:
                timed_kernels["gray_rgb8_to_plane"] = sum(e[0].elapsed_time(e[1]) for e in events) / len(events)
                timed_kernels["sobel3x3_u8_after_gray"] = sum(e[1].elapsed_time(e[2]) for e in events) / len(events)
                region_of["gray_rgb8_to_plane"] = region_of["sobel3x3_u8_after_gray"] = region

        note(ev1, fused, "value", rv1)
        kernel_forms = {"gray_sobel_fused_rgb8" if fused else "sobel3x3_u8_after_gray": lib.gh_last_kernel().decode()}   # which device-kernel form the launch rules picked
        # parity of what was just timed: the WHOLE frame against the compiled reference's SHA-256 (tests/golden/known_answers.json)
        if not args.no_check:
            if is_known_frame and KA:
                want_in, want_edge = KA["frames"]["rand1_16384"]["in"], KA[ka_mode]["rand1_16384"]["edge"]
                got_in = sha256_of(rgb)
                ring_planes = edge_ring if fused else [edge]                  # every plane the timed steps wrote
                got_edges = [sha256_of(e_) for e_ in ring_planes]
                got_edge = got_edges[0] if all(g_ == want_edge for g_ in got_edges) else next(g_ for g_ in got_edges if g_ != want_edge)
                check = f"ok: full-frame SHA-256 of input and of {'all ' + str(len(ring_planes)) + ' edge planes' if len(ring_planes) > 1 else 'the edge plane'} == compiled reference" \
                    if (got_in == want_in and got_edge == want_edge) \
                    else f"MISMATCH (input {'ok' if got_in == want_in else got_in[:16]}, edge {'ok' if got_edge == want_edge else got_edge[:16]})"
            else:
                try:
                    import numpy as np
                    from oracle import pyoracle as O          # checker only, never the thing measured
                    r = min(rows, 96)
                    want = O.sobel(O.gray_to_plane(rgb[0:r + 1].cpu().numpy(), mode))[: r - 1]
                    got = edge[: r - 1].cpu().numpy()
                    check = "ok (first rows vs oracle)" if (np.array_equal(got[1:], want[1:]) and not got[0].any()) else "MISMATCH"
                except Exception as exc:
                    check = f"unavailable ({type(exc).__name__})"
            if check.startswith("MISMATCH"):
                print(f"bench.py: parity check FAILED: {check}", file=sys.stderr)
                return 3

        # second timed region: the other pipeline, same bracket, same K -- so that RGB -> gray and the plane Sobel (or the fused
        # kernel) also have in-region HIP-event figures
        other = None
        if not args.no_kernel_table:
            ev2 = [[torch.cuda.Event(enable_timing=True) for _ in range(n_second)] for _ in range(args.steps)] if fused else None
            rv2 = region_pair()
            run_in(second)
            el2, _ = timed(second, args.steps, ev2, rv2)
            note(ev2, not fused, "second", rv2)
            kernel_forms["sobel3x3_u8_after_gray" if fused else "gray_sobel_fused_rgb8"] = lib.gh_last_kernel().decode()
            other = {"pipeline": "split" if fused else "fused", "ms_per_step": round(el2 / args.steps * 1e3, 4),
                     "Mpixels/s": round(W * H_total * args.steps / el2 / 1e6, 1)}
            if not args.no_check and is_known_frame and KA:
                got_edge = sha256_of(edge)
                if got_edge != KA[ka_mode]["rand1_16384"]["edge"]:
                    print("bench.py: parity check FAILED in the second timed region", file=sys.stderr)
                    return 3
                if fused:            # the second region ran the split pipeline: its luma plane is pinned too
                    if sha256_of(luma[1:rows + 1]) != KA[ka_mode]["rand1_16384"]["gray"]:
                        print("bench.py: luma plane differs from the compiled reference", file=sys.stderr)
                        return 3
                    check += "; split pipeline: luma and edge planes == compiled reference"
                else:
                    check += "; fused pipeline: edge plane == compiled reference"

        # third timed region: the plane Sobel on its own (the reference's sobel_edge is a function of its own too, sobel.c:6-25) -- same
        # bracket, same K, luma planes resident in HBM before the region starts, inputs and outputs rotating over three planes each
        # (a single 268 MB plane re-read launch after launch could be served by the 256 MB Infinity Cache).  Inside the split pipeline
        # above the same kernel is timed between two per-step event records, which cost it ~4 %: both figures are reported.
        sobel_alone = None
        pass_k = None
        if not args.no_kernel_table:
            if not luma[1:rows + 1].any():                                   # (the split pipeline was the FIRST region or did not run)
                chk(lib.gh_gray_rgb8_to_plane_dev(rgb_ptr, luma_ptr + W, rows * W, 3, mode, s_ptr), "gray")
            lumas = [luma, luma.clone(), luma.clone()]
            edges = list(edge_ring[:3]) if len(edge_ring) >= 3 else [edge, torch.empty_like(edge), torch.empty_like(edge)]
            lp, ep = [t.data_ptr() + W for t in lumas], [t.data_ptr() for t in edges]

            def step_sobel(k, ev):
                chk(lib.gh_sobel_u8_rows_dev(lp[k % 3], ep[k % 3], W, rows, 1, 1, s_ptr), "sobel")

            rv3 = region_pair()
            run_in(step_sobel)
            el3, _ = timed(step_sobel, args.steps, None, rv3)
            timed_kernels["sobel3x3_u8"] = rv3[0].elapsed_time(rv3[1]) / args.steps
            kernel_forms["sobel3x3_u8"] = lib.gh_last_kernel().decode()
            region_of["sobel3x3_u8"] = "third"
            sobel_alone = {"pipeline": "plane Sobel alone, inputs / outputs rotating over 3 planes", "ms_per_step": round(el3 / args.steps * 1e3, 4),
                           "Mpixels/s": round(W * H_total * args.steps / el3 / 1e6, 1)}
            if not args.no_check and is_known_frame and KA:
                for e_ in edges:
                    if sha256_of(e_) != KA[ka_mode]["rand1_16384"]["edge"]:
                        print("bench.py: parity check FAILED in the third timed region", file=sys.stderr)
                        return 3
                check += "; plane Sobel alone: all three edge planes == compiled reference"
            del lumas
            # fourth timed region: pass k >= 2 of the reference's multi-pass loop (main_with_sobel.c:51-74 on what the previous pass wrote) as
            # ONE launch -- the plane Sobel with lum(e, e, e) applied on load, 1 R + 1 W byte per pixel; until round 4 a luma pass and a Sobel
            # pass (4 B/px, two launches).  Inputs: the three pass-1 edge planes just checked; outputs rotate over three planes as well.
            outs = [torch.empty_like(edge) for _ in range(3)]
            ip, op = [t.data_ptr() for t in edges], [t.data_ptr() for t in outs]

            def step_pass_k(k, ev):
                chk(lib.gh_sobel_of_gray_u8_dev(ip[k % 3], op[k % 3], W, rows, mode, s_ptr), "sobel_of_gray")

            rv4 = region_pair()
            run_in(step_pass_k)
            el4, _ = timed(step_pass_k, args.steps, None, rv4)
            timed_kernels["pipeline_pass_k"] = rv4[0].elapsed_time(rv4[1]) / args.steps
            kernel_forms["pipeline_pass_k"] = lib.gh_last_kernel().decode()
            region_of["pipeline_pass_k"] = "fourth"
            pass_k = {"pipeline": "pass k >= 2 of the gray+Sobel loop in one launch (gh_sobel_of_gray_u8_dev), inputs / outputs rotating over 3 planes",
                      "ms_per_step": round(el4 / args.steps * 1e3, 4), "Mpixels/s": round(W * H_total * args.steps / el4 / 1e6, 1)}
            if not args.no_check:
                # the launch it replaces: gray_of_plane (lum(e,e,e) plane) then the plane Sobel -- both pinned above; bit for bit
                tmp = torch.empty_like(edge)
                chk(lib.gh_gray_of_plane_dev(ip[0], tmp.data_ptr(), rows * W, mode, s_ptr), "gray_of_plane")
                ref2 = torch.empty_like(edge)
                chk(lib.gh_sobel_u8_rows_dev(tmp.data_ptr(), ref2.data_ptr(), W, rows, 1, 1, s_ptr), "sobel")
                torch.cuda.synchronize()
                if not all(bool(torch.equal(o_, ref2)) for o_ in outs):
                    print("bench.py: parity check FAILED in the fourth timed region (pass k in one launch != gray_of_plane + Sobel)", file=sys.stderr)
                    return 3
                check += "; pass k in one launch == gray_of_plane + Sobel"
                del tmp, ref2
            del edges, outs

        # outside the timed regions: the in-place reference-ABI kernel, and the copy ceiling of this GPU
        def time_kernel(fn, reps=30):
            for _ in range(150):      # run-in: a change of kernel is followed by the same clock dip as a start from idle
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(reps):
                fn()
            b.record(stream)
            b.synchronize()
            return a.elapsed_time(b) / reps

        kern_ms = dict(timed_kernels)
        copy_ceiling = None
        if not args.no_kernel_table:
            scratch = rgb.clone()
            kern_ms["gray_inplace_rgb8"] = time_kernel(lambda: api.gray_inplace(scratch, mode))
            region_of["gray_inplace_rgb8"] = None
            nb = rgb.numel()
            sp, dp = rgb.data_ptr(), scratch.data_ptr()
            ms = time_kernel(lambda: chk(lib.gh_copy_stream_dev(sp, dp, nb, s_ptr), "copy"))
            copy_ceiling = 2 * nb / (ms * 1e-3) / 1e9
            del scratch

        value = W * H_total * args.steps / elapsed / 1e6
        kern_table = {}
        for name, ms in kern_ms.items():
            gbs = ALGO_BYTES_PER_PX[name] * px_rank / (ms * 1e-3) / 1e9
            kern_table[name] = {"ms": round(ms, 4), "algo_bytes": ALGO_BYTES_PER_PX[name] * px_rank,
                                "GB/s": round(gbs, 1), "frac_of_8TBs": round(gbs / HBM_PEAK_GBS, 4),
                                "frac_of_copy_ceiling": round(gbs / copy_ceiling, 4) if copy_ceiling else None,
                                "Mpixels/s": round(px_rank / (ms * 1e-3) / 1e6, 1),
                                "in_timed_region": region_of.get(name) is not None, "timed_region": region_of.get(name)}
            if name in kernel_forms:
                kern_table[name]["device_kernel"] = kernel_forms[name]
        first_kernels = {k: v for k, v in timed_kernels.items() if region_of[k] == "value"}
        dom = max(first_kernels, key=first_kernels.get)
        a = kern_table[dom]["GB/s"]
        traffic, traffic_src, traffic_cross = measured_traffic(dom, px_rank, kern_table[dom]["ms"], kern_table[dom].get("device_kernel", ""))
        if traffic_src:
            traffic_src += " (FETCH_SIZE / WRITE_SIZE count the L2's fabric-side requests, Infinity-Cache hits included -- not DRAM bytes alone; the region "
            traffic_src += f"re-reads one 805 MB frame and writes {n_ring} edge plane(s) in rotation, all with nontemporal accesses)"
        roofline = {"bound": "hbm", "kernel": dom, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(a / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src, "traffic_cross_check": traffic_cross,
                    "output_planes_in_rotation": n_ring if dom == "gray_sobel_fused_rgb8" else 1,
                    "algo_bytes_per_launch": kern_table[dom]["algo_bytes"], "avg_launch_ms": kern_table[dom]["ms"],
                    "copy_ceiling_GBs": round(copy_ceiling, 1) if copy_ceiling else None,
                    "frac_of_copy_ceiling": round(a / copy_ceiling, 4) if copy_ceiling else None}
        out.update({"value": round(value, 1), "ms_per_step": round(elapsed / args.steps * 1e3, 4), "roofline": roofline,
                    "kernels": kern_table, "ms_per_step_events": round(step_ms_events, 4),
                    "repeats": {"ms_per_step": [round(v, 4) for v in rep_ms], "min": round(min(rep_ms), 4),
                                "median": round(sorted(rep_ms)[len(rep_ms) // 2], 4), "max": round(max(rep_ms), 4),
                                "note": "the K-step bracket run again back to back; value / ms_per_step are the first entry"},
                    "second_timed_region": other, "third_timed_region": sobel_alone, "fourth_timed_region": pass_k,
                    "copy_ceiling_GBs": round(copy_ceiling, 1) if copy_ceiling else None,
                    "host_enqueue_us_per_step": round(host_enqueue * 1e6 / args.steps, 1)})
        parallelism = "single GPU"
    else:
        # ================================ N GPUs, row shards ===============================================================
        def torch_step(k):            # exchange issued by torch.distributed P2P ops, frame k % NBUF, result in that runner's edge
            r = runners[k % NBUF]
            return r.step() if fused else r.step(rgbs[k % NBUF])

        pipelined_impl, edge_of = torch_step, (lambda k: runners[k % NBUF].edge)
        single_impl, single_edge = (lambda k: torch_step(0)), runners[0].edge
        stream_impl = torch_step          # a stream of DIFFERENT frames, each step the whole single-frame step on the next one
        batched_impl = None               # (native only) the same stream with the halo rows of four future frames per exchange
        wd.arm("validation of libgrayhip's sharded steps against torch.distributed on the first frames", 300 * budget_scale)
        if native is not None:
            # libgrayhip issues the exchange itself: halo of frame k+2 beside the kernel(s) of frame k, one C call per step
            AHEAD = 2
            edge_native = torch.empty((rows, W), dtype=torch.uint8, device=device)
            try:
                if fused:
                    # stream of frames: the halo rows of frame k + 1 travel beside the launch on frame k, the two sides meeting through the
                    # polled dwords of the halo-flag form (gh_sharded_fused_stream_step_dev)
                    pstep = native.bind_fused_stream_step(halo_bufs, edge_native, rows, mode)
                    native_step = pstep
                    sstep = native.bind_fused_step(halo_bufs[0], edge_native, rows, mode)
                    native_single = lambda k: sstep()
                    ring_steps = [native.bind_fused_step(hb, edge_native, rows, mode) for hb in halo_bufs]
                    native_stream = lambda k: ring_steps[k % NBUF]()
                    # batched form: the halos of BATCH future frames in one group, a step = one launch (checked below like the other)
                    bstep = native.bind_fused_step_batched(halo_bufs, edge_native, rows, mode, BATCH) if args.halo_batch else None
                else:
                    pstep = native.bind_sobel_stream_step(halo_bufs, edge_native, rows)
                    grays = [backend.bind_gray_to_plane(rgbs[i], halo_bufs[i][1:rows + 1]) for i in range(NBUF)]
                    for g_ in grays:
                        g_()          # prologue: luma of every frame of the ring exists before its halo is first requested

                    def native_step(k):
                        grays[(k + 1) % NBUF]()          # luma of the frame whose halo rows this step sends on their way
                        pstep(k)
                    sstep = native.bind_sobel_step(halo_bufs[0], edge_native, rows)

                    def native_single(k):
                        grays[0]()
                        sstep()
                    ring_steps = [native.bind_sobel_step(hb, edge_native, rows) for hb in halo_bufs]

                    def native_stream(k):
                        grays[k % NBUF]()
                        ring_steps[k % NBUF]()
                # every frame of the ring through both paths: the C++ path must reproduce torch.distributed bit for bit
                same = True
                for k in range(NBUF):
                    torch_step(k)
                torch.cuda.synchronize()
                for k in range(2 * NBUF):
                    native_step(k)
                    torch.cuda.synchronize()
                    same = same and bool(torch.equal(edge_native, runners[k % NBUF].edge))
                # the single-frame step (`value`): libgrayhip's default form (halo flag) three times over; should it disagree with
                # torch.distributed on any rank, every rank switches to the serial form (exchange, then launch) and checks again
                def single_agrees():
                    good = True
                    try:
                        for _ in range(3):
                            native_single(0)
                            torch.cuda.synchronize()
                            good = good and bool(torch.equal(edge_native, runners[0].edge)) and native.halo_timeouts() == 0
                    except Exception as exc_:
                        print(f"bench.py: rank {rank}: single-frame step failed: {exc_}", file=sys.stderr)
                        good = False
                    return all_ranks_agree(good or rehearsal_loopback)

                step_form = int(os.environ.get("GH_SHARD_STEP_FORM", "2"))
                single_ok = single_agrees()
                if not single_ok and step_form != 1:
                    native.set_step_form(1)
                    step_form = 1
                    single_ok = single_agrees()
                    exchange_note = "the halo-flag form of the single-frame step disagreed with torch.distributed; serial form (exchange, then launch) in its place"
                pipe_ok, same = same, single_ok          # `same` decides whether the native path is adopted at all: the single-frame step (`value`)
                same_b = False
                if same and fused and bstep is not None:
                    same_b = True
                    for k in range(2 * NBUF + BATCH):
                        bstep(k)
                        torch.cuda.synchronize()
                        same_b = same_b and bool(torch.equal(edge_native, runners[k % NBUF].edge))
            except Exception as exc:
                same, same_b, pipe_ok, exchange_note = False, False, False, f"native step failed on rank {rank}: {exc}"
            if all_ranks_agree(same or rehearsal_loopback):
                exchange = ("libgrayhip RCCL communicator (csrc/comm.hip): ncclSend/ncclRecv group on a side stream, issued from C++; "
                            + "pipelined form: the halo rows of frame k+1 travel beside the launch on frame k, no event between the streams")
                single_impl, single_edge = native_single, edge_native
                stream_impl = native_stream
                edge_of = lambda k: edge_native
                if all_ranks_agree(pipe_ok or rehearsal_loopback):
                    pipelined_impl = native_step
                else:                              # the stream step disagreed: the checked single-frame step on every frame in its place
                    pipelined_impl = native_stream
                    exchange_note = exchange_note or "the pipelined stream step disagreed with torch.distributed on the first frames; the single-frame step per frame in its place"
                if fused and bstep is not None and all_ranks_agree(same_b or rehearsal_loopback):
                    batched_impl = bstep          # (k runs on from 0 through run-in, warm-up and timed steps: a batch request every BATCH-th step)
                elif fused and bstep is not None:
                    exchange_note = exchange_note or "batched halo prefetch disagreed with torch.distributed on the first frames: not timed"
            else:
                exchange_note = exchange_note or "native exchange disagreed with torch.distributed on the first frames; using torch"
                native = None

        wd.disarm()
        calls = {}

        def with_events(impl, name):
            calls[name] = 0

            def fn(_k, ev):          # the step number runs on through run-in, warm-up and timed steps: the frame ring never skips
                k = calls[name]
                calls[name] = k + 1
                impl(k)
            return fn

        def run_form(impl, k_steps, name):
            # one HIP-event pair around the K steps (per-step pairs cost every step ~5 us of GPU time: 12 % of a 40 us sharded step)
            pair = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
            fn = with_events(impl, name)
            run_in(fn)
            el, host = timed(fn, k_steps, None, pair)
            return {"ms_per_step": round(el / k_steps * 1e3, 4), "Mpixels/s": round(W * H_total * k_steps / el / 1e6, 1),
                    "ms_per_step_events_rank0": round(pair[0].elapsed_time(pair[1]) / k_steps, 4),
                    "host_enqueue_us_per_step": round(host * 1e6 / k_steps, 1), "steps": k_steps}, el

        forms = {}
        xc0 = native.exchange_count() if native is not None else 0
        # (a) ONE frame, strong-scaled (BASELINE configs[3]): every step = one halo exchange + the shard -- `value`
        with phase("single_frame (value)", 180):
            t_form = time.perf_counter()
            forms["single_frame"], elapsed = run_form(single_impl, args.steps, "single_frame")
            value = W * H_total * args.steps / elapsed / 1e6
            step_ms_events = forms["single_frame"]["ms_per_step_events_rank0"]
            rep_ms = [elapsed / args.steps * 1e3]
            fn_single = with_events(single_impl, "single_frame_repeats")
            for _ in range(max(args.repeats, 1) - 1):
                el_r, _ = timed(fn_single, args.steps, None, None)
                rep_ms.append(el_r / args.steps * 1e3)
            if native is not None:
                forms["single_frame"]["exchanges_per_step"] = round((native.exchange_count() - xc0) / max(calls["single_frame"] + calls["single_frame_repeats"], 1), 3)
            forms["single_frame"]["wall_s"] = round(time.perf_counter() - t_form, 2)
            # Sum of the edge plane over all ranks against the compiled reference's known answer (single-frame form, frame rand(seed=1))
            edge_sum_check = "skipped"
            if not args.no_check and not rehearsal_loopback:
                tsum = single_edge.sum(dtype=torch.int64).reshape(1)
                if rehearsal:
                    tsum = tsum.cpu()
                dist.all_reduce(tsum)
                if is_known_frame and KA and args.scaling == "strong" and ka_mode == "IEEE":
                    want = int(KA["IEEE"]["rand1_16384"]["edge_sum"])
                    edge_sum_check = "ok: sum of the sharded edge plane == compiled reference" if int(tsum.item()) == want else \
                        f"MISMATCH ({int(tsum.item())} != {want})"
                else:
                    edge_sum_check = f"unpinned geometry (sum {int(tsum.item())})"
            halo_timeouts = native.halo_timeouts() if native is not None else 0
            timeouts_ok = all_ranks_agree(halo_timeouts == 0)
        # From here on `value` exists: whatever happens to the optional forms below, the line is written with it (the watchdog holds a copy).
        value_good = not edge_sum_check.startswith("MISMATCH") and timeouts_ok
        if wd is not None:
            wd.partial.update({"value": round(value, 1), "ms_per_step": round(elapsed / args.steps * 1e3, 4), "step_forms": forms,
                               "edge_sum_check": edge_sum_check, "halo_flag_timeouts": halo_timeouts,
                               "config": {"workload": f"{W}x{H_total} RGB8 frame rand(seed=1) row-sharded x{world} (BASELINE configs[3]), pipeline={args.pipeline}, mode={args.mode}"},
                               "incomplete": "the optional step forms did not all finish"})
            wd.good = value_good

        inject = os.environ.get("GH_BENCH_INJECT", "")

        def guarded(name, seconds, body):
            """An optional figure: a failure -- on any rank -- lands in the line as <name>.error instead of costing the run its `value`."""
            t0_ = time.perf_counter()
            res, err = None, None
            with phase(name, seconds):
                try:
                    if inject == f"fail:{name}":               # (tests: a form that throws / never returns must not cost the line)
                        raise RuntimeError("failure injected by GH_BENCH_INJECT")
                    if inject == f"hang:{name}":
                        time.sleep(1e6)
                    res = body()
                except Exception as exc:
                    err = f"{type(exc).__name__}: {exc}"
                    print(f"bench.py: rank {rank}: {name} failed: {err}", file=sys.stderr)
                everyone = all_ranks_agree(err is None)
            if not everyone:
                res = {"error": err or "failed on another rank"}
            if isinstance(res, dict):
                res["wall_s"] = round(time.perf_counter() - t0_, 2)
            return res

        # Per-rank breakdown of the single-frame step (HIP events on every rank, same K everywhere): the shard kernel alone (no
        # exchange), the exchange alone (nothing beside it), the whole step.  step - kernel = what the exchange costs on top of the
        # kernel after overlap; kernel differences between ranks = the edge ranks' missing halo row or a slow GPU.
        def rank_events(fn, k_steps):
            for _ in range(8):
                fn()
            fence()
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record(stream)
            for _ in range(k_steps):
                fn()
            b_.record(stream)
            fence()
            return a_.elapsed_time(b_) / k_steps * 1e3           # us

        def breakdown_body():
            kb = max(args.steps, 20)
            hb0 = halo_bufs[0]
            if fused:
                kern_only = backend.bind_fused_rows(hb0, single_edge, rows, geom.top_is_border, geom.bottom_is_border)
            else:
                kern_only = backend.bind_sobel_rows(hb0, single_edge, rows, geom.top_is_border, geom.bottom_is_border)
            us_kernel = rank_events(kern_only, kb)
            us_exchange = None
            if native is not None and not rehearsal_loopback:
                up_p, dn_p = (rank - 1 if rank > 0 else -1), (rank + 1 if rank < world - 1 else -1)
                nb_row = hb0[0].numel()
                xfn = lambda: native.halo_exchange(up_p, hb0[1], hb0[0], dn_p, hb0[rows], hb0[rows + 1], nb_row, stream.cuda_stream)
                us_exchange = rank_events(xfn, kb)
            cnt_b = [0]

            def step_b():
                single_impl(cnt_b[0])
                cnt_b[0] += 1
            us_step = rank_events(step_b, kb)
            mine = torch.tensor([us_kernel, us_exchange if us_exchange is not None else -1.0, us_step], dtype=torch.float64, device=device)
            allr = [torch.empty_like(mine) for _ in range(world)]
            if rehearsal:
                allr = [t_.cpu() for t_ in allr]
                dist.all_gather(allr, mine.cpu())
            else:
                dist.all_gather(allr, mine)
            rows_ = [[round(float(v), 1) for v in t_.tolist()] for t_ in allr]
            return {"unit": "us per step, HIP events on each rank's stream", "steps": kb,
                    "shard_kernel_alone": [r_[0] for r_ in rows_],
                    "exchange_alone": [r_[1] if r_[1] >= 0 else None for r_ in rows_],
                    "single_frame_step": [r_[2] for r_ in rows_],
                    "exchange_not_hidden": [round(r_[2] - r_[0], 1) for r_ in rows_],
                    "note": "step - kernel = what the halo exchange costs after overlap (rank 0 and the last rank wait for ONE halo row and keep the frame's "
                            "zero border on the other side); exchange_alone is the unloaded RCCL group (beside a bandwidth-bound kernel the same group "
                            "takes ~3x as long: DESIGN.md section 6)"}

        breakdown = guarded("per_rank breakdown", 120, breakdown_body)

        # every rank's halo rows must hold its neighbours' edge rows: one more single-frame step on frame 0, then what its exchange delivered
        def halo_rows_body():
            good = True
            single_impl(0)
            torch.cuda.synchronize()
            lrgb, hbuf = rgbs[0], halo_bufs[0]
            firsts = [torch.empty((W, 3), dtype=torch.uint8, device=device) for _ in range(world)]
            lasts = [torch.empty((W, 3), dtype=torch.uint8, device=device) for _ in range(world)]
            if rehearsal:
                firsts, lasts = [t_.cpu() for t_ in firsts], [t_.cpu() for t_ in lasts]
                dist.all_gather(firsts, lrgb[0].contiguous().cpu())
                dist.all_gather(lasts, lrgb[rows - 1].contiguous().cpu())
                firsts, lasts = [t_.to(device) for t_ in firsts], [t_.to(device) for t_ in lasts]
            else:
                dist.all_gather(firsts, lrgb[0].contiguous())
                dist.all_gather(lasts, lrgb[rows - 1].contiguous())
            up_row, down_row = hbuf[0], hbuf[rows + 1]
            conv = (lambda t: t) if fused else (lambda t: api.gray_to_plane(t.reshape(1, W, 3).contiguous(), mode).reshape(W))
            if rank > 0:
                good &= bool(torch.equal(up_row, conv(lasts[rank - 1])))
            if rank < world - 1:
                good &= bool(torch.equal(down_row, conv(firsts[rank + 1])))
            return {"ok": bool(good or rehearsal_loopback)}

        hr = guarded("halo rows check", 90, halo_rows_body)
        halo_ok = all_ranks_agree(bool(hr.get("ok", False)) or "error" in hr) if isinstance(hr, dict) else False
        halo_rows_note = hr.get("error") if isinstance(hr, dict) else None
        if native is not None:
            halo_timeouts = native.halo_timeouts()
        if not all_ranks_agree(halo_timeouts == 0):
            halo_ok = False

        if not args.no_check and rank == 0 and not rehearsal_loopback:
            try:
                import numpy as np
                from oracle import pyoracle as O          # checker only
                lrgb, ledge, hbuf = rgbs[0], single_edge, halo_bufs[0]
                r = min(rows, 96)
                want = O.sobel(O.gray_to_plane(lrgb[0:r + 1].cpu().numpy(), mode))[: r - 1]
                got = ledge[: r - 1].cpu().numpy()
                good = np.array_equal(got[1:], want[1:]) and not got[0].any()
                if fused and rows > 40:
                    # the shard's last rows, which depend on the halo row received from rank 1
                    src = hbuf[rows - 30:rows + 2].cpu().numpy()                # buffer rows = shard rows rows-31 .. rows-1 + the halo row
                    want = O.sobel(O.gray_to_plane(src, mode))[1:-1]             # shard rows rows-30 .. rows-1
                    good = good and np.array_equal(ledge[rows - 30:rows].cpu().numpy(), want)
                check = "ok (rank 0, single-frame step: first rows and the rows behind the received halo vs oracle)" if good else "MISMATCH"
            except Exception as exc:  # the checker is optional on the GPU box; a mismatch is never hidden
                check = f"unavailable ({type(exc).__name__})"
        bad = check == "MISMATCH" or not halo_ok or edge_sum_check.startswith("MISMATCH")
        if all_ranks_agree(not bad) is False:
            bad = True
        if bad:
            print(f"bench.py: parity check FAILED (check={check}, halo rows ok={halo_ok}, halo-flag timeouts={halo_timeouts}, edge sum: {edge_sum_check})",
                  file=sys.stderr)
            if wd is not None:
                wd.finish()
            return 3

        # ---- the optional forms, each guarded: a failure costs its own figure only ----------------------------------------------------
        def form_body(impl, k_steps, name, count_exchanges=False):
            def body():
                x0 = native.exchange_count() if (native is not None and count_exchanges) else 0
                res, _ = run_form(impl, k_steps, name)
                if native is not None and count_exchanges:
                    res["exchanges_per_step"] = round((native.exchange_count() - x0) / max(calls[name], 1), 3)
                if native is not None:
                    native.check()                                  # (behind run_form's final synchronisation)
                return res
            return body

        # (a') the single-frame step with the SERIAL form forced on every rank (exchange on the caller's stream, then one launch): the comparison
        # the halo-flag form has to win -- over xGMI an exchange alone may cost more than the loopback's 8-11 us, and this line then says by how much
        if native is not None:
            def serial_body():
                native.set_step_form(1)
                try:
                    return form_body(single_impl, args.steps, "single_frame_serial_form", True)()
                finally:
                    native.set_step_form(-1)
            forms["single_frame_serial_form"] = guarded("single_frame_serial_form", 120, serial_body)

        # (b) end to end: the shard comes from pinned host memory first (H2D scatter), then the single-frame step
        def e2e_body():
            host_shard = torch.empty((rows, W, 3), dtype=torch.uint8, pin_memory=True)
            host_shard.copy_(rgbs[0])
            torch.cuda.synchronize()

            def e2e_impl(k):
                rgbs[0].copy_(host_shard, non_blocking=True)
                single_impl(k)
            return form_body(e2e_impl, max(1, min(args.steps, 20)), "end_to_end_h2d")()

        forms["end_to_end_h2d"] = guarded("end_to_end_h2d", 120, e2e_body)
        # (c) a stream of DIFFERENT frames (ring of eight), every step the whole single-frame step -- its own exchange and its own launch --
        # on the next frame: what `value` becomes when no frame is seen twice (nothing of it can sit in the 256 MB Infinity Cache)
        forms["stream_of_frames"] = guarded("stream_of_frames", 120, form_body(stream_impl, args.steps, "stream_of_frames"))
        # (d) the same stream with the halo rows of four future frames per exchange (events; round 2's form): fewer, larger exchanges
        if batched_impl is not None:
            forms["pipelined_batched"] = guarded("pipelined_batched", 120, form_body(batched_impl, args.steps, "pipelined_batched", True))
        # (e) the same stream software-pipelined: the halo rows of the next frame travel beside this frame's launch
        forms["pipelined"] = guarded("pipelined", 120, form_body(pipelined_impl, args.steps, "pipelined", True))
        if native is not None:
            halo_timeouts = native.halo_timeouts()
        # what the communicator of every rank decided and saw: the form its single-frame steps take now (the halo-flag form reviews itself and
        # goes back to the serial form where the exchange does not run beside the launch), its edge bands' average standstill, an exchange alone
        comm_diag = None
        if native is not None:
            def diag_body():
                w_, us_ = native.halo_wait_stats()
                mine = torch.tensor([float(native.step_form_in_use()), us_ / max(w_, 1), native.exchange_alone_us(), float(w_), float(native.side_stream_replacements())],
                                    dtype=torch.float64, device=device)
                allr = [torch.empty_like(mine) for _ in range(world)]
                if rehearsal:
                    allr = [t_.cpu() for t_ in allr]
                    dist.all_gather(allr, mine.cpu())
                else:
                    dist.all_gather(allr, mine)
                rows_ = [t_.tolist() for t_ in allr]
                return {"step_form_in_use": [int(r_[0]) for r_ in rows_], "edge_band_standstill_avg_us": [round(r_[1], 1) for r_ in rows_],
                        "exchange_alone_at_init_us": [round(r_[2], 1) for r_ in rows_], "standstill_reports": [int(r_[3]) for r_ in rows_],
                        "side_stream_replacements": [int(r_[4]) for r_ in rows_],
                        "note": "form 2 = halo flag, 1 = serial; when its edge bands stand still longer than an exchange alone would take a rank swaps its side stream (another hardware "
                                "queue) up to twice, then goes to form 1 by itself"}
            comm_diag = guarded("communicator diagnostics", 60, diag_body)
            if isinstance(comm_diag, dict) and "step_form_in_use" in comm_diag:
                step_form = min(comm_diag["step_form_in_use"]) if len(set(comm_diag["step_form_in_use"])) == 1 else step_form

        bpp = 4 if fused else 6
        a = bpp * px_rank / (step_ms_events * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": ("gray_sobel_fused_rgb8" if fused else "gray_rgb8_to_plane+sobel3x3_u8")
                    + " (per rank, single-frame step incl. halo exchange)", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(a / HBM_PEAK_GBS, 4), "traffic": None,
                    "traffic_source": "not recorded for shards (the 1-GPU line carries the PMC figure of the same kernel)"}
        out.update({"value": round(value, 1), "ms_per_step": round(elapsed / args.steps * 1e3, 4), "roofline": roofline,
                    "step_single_frame_ms": forms["single_frame"]["ms_per_step"], "step_pipelined_ms": forms["pipelined"].get("ms_per_step"),
                    "step_end_to_end_ms": forms["end_to_end_h2d"].get("ms_per_step"), "step_stream_of_frames_ms": forms["stream_of_frames"].get("ms_per_step"),
                    "step_forms": forms,
                    "wall_clock_s": dict(wd.spent) if wd is not None else None,
                    "working_set_note": (f"per GPU the step touches {px_rank * 4 / 1e6:.0f} MB (shard in, edge plane out); the single-frame form re-reads ONE shard, of which a "
                                         "256 MB Infinity Cache can keep a part once that is small enough (N >= 4 for this frame) -- stream_of_frames is the same step "
                                         "on eight different frames, where nothing is seen twice"),
                    "repeats": {"ms_per_step": [round(v, 4) for v in rep_ms], "min": round(min(rep_ms), 4),
                                "median": round(sorted(rep_ms)[len(rep_ms) // 2], 4), "max": round(max(rep_ms), 4),
                                "note": "the K-step bracket of the single-frame form run again back to back; value / ms_per_step are the first entry"},
                    "per_rank": breakdown, "communicators": comm_diag, "exchange_fallback": native is None and args.exchange == "native",
                    "exchange_issuer": "libgrayhip (csrc/comm.hip)" if native is not None else "torch.distributed", "halo_flag_timeouts": halo_timeouts,
                    "shard_step_form": ({2: "halo flag (exchange on a side stream beside one launch; edge bands wait for a flag)", 1: "serial (exchange, then one launch)",
                                         0: "events (exchange || interior rows, then edge rows)"}.get(step_form) if native is not None else None),
                    "ms_per_step_events": step_ms_events, "edge_sum_check": edge_sum_check,
                    "host_enqueue_us_per_step": forms["single_frame"]["host_enqueue_us_per_step"]})
        halo = "RGB row (the fused kernel recomputes halo luma)" if fused else "luma row"
        parallelism = (f"row-sharded x{world}; per step one {halo} is exchanged with each neighbour via RCCL send/recv; value = single-frame form "
                       f"(ONE frame split {world} ways, BASELINE configs[3]); end_to_end_h2d and pipelined (stream of frames, ring of {NBUF}) in step_forms")

    if rank == 0:
        final = {
            "metric": METRIC,
            "value": out.pop("value"),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": out.pop("ms_per_step"),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"{W}x{H_total} RGB8 frame rand(seed=1) (SURVEY 8c generator) resident in HBM -> luma -> 3x3 Sobel edge plane "
                            f"(BASELINE configs[{2 if world == 1 else 3}]), pipeline={args.pipeline}, mode={args.mode}",
                "width": W, "height": H_total, "rows_per_gpu": rows,
                "parallelism": parallelism,
                "halo_exchange": exchange + (f" [{exchange_note}]" if exchange_note else ""),
            },
            "device": device_topology,
            "ranks": ranks_met,
            "rccl_ranks": rccl_ranks,
            "exchange_fallback": False,
        }
        final.update(out)
        final.update({"parity_check": check, "halo_rows_check": ("ok" if halo_ok else "MISMATCH") if world == 1 or not halo_rows_note else f"not run ({halo_rows_note})",
                      "settle_ms": args.settle_ms})
        if rehearsal:
            final["rehearsal"] = "ranks share one GPU over gloo: code-path rehearsal only, not a measurement"
        if world == 1 and not args.no_cpu_baseline:
            try:
                final["cpu_baseline"] = cpu_baseline(W, H_total, args.cpu_budget_s, topo)
            except Exception as exc:
                final["cpu_baseline"] = {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "unavailable",
                                         "sample": f"{type(exc).__name__}: {exc}"}
        sys.stdout.flush()
        if wd is not None:
            wd.line_written()          # (the line is complete: the watchdog must not write a second one)
        os.write(json_fd, (json.dumps(final) + "\n").encode())

    if world > 1:
        wd.arm("shutdown (barrier, communicator and process-group teardown)", 120 * budget_scale)
        torch.cuda.synchronize()
        dist.barrier()
        for comm in comms_to_close:        # every rank has finished with the communicator: release it before the process group
            try:
                comm.close()
            except Exception:
                pass
        dist.destroy_process_group()
        wd.finish()
    return 0


if __name__ == "__main__":
    sys.exit(main())
