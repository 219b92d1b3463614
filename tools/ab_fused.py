#!/usr/bin/env python3
"""Interleaved A/B timing of one C-ABI kernel under several tuning-knob settings, inside one process (developer tool).

usage: ab_fused.py [--op fused|sobel|gray] [--w W] [--h H] [--rounds N] "NAME=K1:V1,K2:V2" ...
Each variant is a set of environment knobs (GH_TUNE_DYNAMIC makes the library re-read them per launch).
Runs against the probe build (`make probes` -> build_probes/libgrayhip_probes.so): the release library reads no tuning knob.
"""
import argparse, importlib, os, sys, hashlib, statistics
os.environ["GH_TUNE_DYNAMIC"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the knobs exist in the probe build only (`make probes`); GRAYHIP_LIB=<path> to time another build
os.environ.setdefault("GRAYHIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_probes", "libgrayhip_probes.so"))
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--op", default="fused"); ap.add_argument("--w", type=int, default=16384); ap.add_argument("--h", type=int, default=0)
ap.add_argument("--rounds", type=int, default=9); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--in-offset-rows", type=int, default=0, help="input frame starts this many rows into its allocation")
ap.add_argument("--out-offset-rows", type=int, default=0, help="output plane starts this many rows into its allocation")
ap.add_argument("--out-ring", type=int, default=1, help="rotate the output over this many buffers (a single 268 MB output plane fits the 256 MB Infinity Cache: write-back stores can then look faster than HBM allows)")
ap.add_argument("--in-ring", type=int, default=1, help="rotate the input over this many frames (defeats any Infinity Cache read hits between launches)")
ap.add_argument("variants", nargs="*")
a = ap.parse_args()
W, H = a.w, (a.h or a.w)
pkg = importlib.import_module("openmp-parallel-computing_amd"); api = pkg.hostapi; api.init(0)
torch.manual_seed(1)
io, oo = a.in_offset_rows, a.out_offset_rows
rgb = torch.randint(0, 256, (H + io, W, 3), dtype=torch.uint8, device="cuda")[io:]
plane = torch.randint(0, 256, (H + io, W), dtype=torch.uint8, device="cuda")[io:]
out = torch.empty((H + oo, W), dtype=torch.uint8, device="cuda")[oo:]
rgba = torch.randint(0, 256, (H, W, 4), dtype=torch.uint8, device="cuda") if a.op in ("fused4", "gray4", "inplace4") else None
out3 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda") if a.op in ("conv", "minmax") else None
scr = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda") if a.op in ("inplace", "replicate") else None
mm = torch.tensor([256, -1], dtype=torch.int32, device="cuda")
sums = torch.zeros(3, dtype=torch.int64, device="cuda")
ops = {"fused": (lambda: api.gray_sobel_fused(rgb, 0, out=out), 4), "sobel": (lambda: api.sobel(plane, out=out), 2),
       "gray": (lambda: api.gray_to_plane(rgb, 0, out=out), 4),
       "inplace": (lambda: api.gray_inplace(scr, 0), 6),
       "extract": (lambda: api.extract_plane(rgb, out=out), 4),
       "replicate": (lambda: api.replicate_plane(plane, scr), 4),
       "grayplane": (lambda: api.gray_of_plane(plane, 0, out=out), 2),
       "fused4": (lambda: api.gray_sobel_fused(rgba, 0, out=out), 5),
       "gray4": (lambda: api.gray_to_plane(rgba, 0, out=out), 5),
       "inplace4": (lambda: api.gray_inplace(rgba, 0), 8),
       "conv": (lambda: api.conv3x3(rgb, [1, 2, 1, 2, 4, 2, 1, 2, 1], 16, out=out3), 6),
       "minmax": (lambda: pkg.abi.check(pkg.abi.load().gh_mean_gray_minmax_dev(rgb.data_ptr(), out3.data_ptr(), W * H, mm.data_ptr(), torch.cuda.current_stream().cuda_stream)), 6),
       "sum": (lambda: pkg.abi.check(pkg.abi.load().gh_sum_rgb8_dev(rgb.data_ptr(), W * H, sums.data_ptr(), torch.cuda.current_stream().cuda_stream)), 3)}
fn, bpp = ops[a.op]
if a.in_ring > 1 and a.op in ("sum", "minmax"):
    rgbs = [rgb] + [rgb.clone() for _ in range(a.in_ring - 1)]
    cnt = [0]
    L = pkg.abi.load()
    def fn():
        cnt[0] += 1
        r = rgbs[cnt[0] % len(rgbs)]
        if a.op == "sum": pkg.abi.check(L.gh_sum_rgb8_dev(r.data_ptr(), W * H, sums.data_ptr(), torch.cuda.current_stream().cuda_stream))
        else: pkg.abi.check(L.gh_mean_gray_minmax_dev(r.data_ptr(), out3.data_ptr(), W * H, mm.data_ptr(), torch.cuda.current_stream().cuda_stream))
if (a.out_ring > 1 or a.in_ring > 1) and a.op in ("fused", "sobel", "gray", "extract", "grayplane"):
    outs = [out] + [torch.empty((H, W), dtype=torch.uint8, device="cuda") for _ in range(a.out_ring - 1)]
    rgbs = [rgb] + [rgb.clone() for _ in range(a.in_ring - 1)] if a.op in ("fused", "gray", "extract") else [rgb]
    planes = [plane] + [plane.clone() for _ in range(a.in_ring - 1)] if a.op in ("sobel", "grayplane") else [plane]
    cnt = [0]
    base_ops = {"fused": lambda i, o: api.gray_sobel_fused(rgbs[i % len(rgbs)], 0, out=o), "sobel": lambda i, o: api.sobel(planes[i % len(planes)], out=o),
                "gray": lambda i, o: api.gray_to_plane(rgbs[i % len(rgbs)], 0, out=o), "extract": lambda i, o: api.extract_plane(rgbs[i % len(rgbs)], out=o),
                "grayplane": lambda i, o: api.gray_of_plane(planes[i % len(planes)], 0, out=o)}[a.op]
    def fn():
        cnt[0] += 1
        base_ops(cnt[0], outs[cnt[0] % len(outs)])
variants = [("base", {})]
for v in a.variants:
    name, _, kv = v.partition("=")
    variants.append((name, dict(x.split(":") for x in kv.split(",") if x)))
allkeys = {k for _, d in variants for k in d}
def setenv(d):
    for k in allkeys: os.environ.pop(k, None)
    os.environ.update(d)
times = {n: [] for n, _ in variants}; shas = {}
for rnd in range(a.rounds + 1):
    for n, d in variants:
        setenv(d)
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [fn() for _ in range(a.reps)]; e1.record(); e1.synchronize()
        if rnd: times[n].append(e0.elapsed_time(e1) / a.reps)
        else: shas[n] = hashlib.sha256((out3 if out3 is not None else out).cpu().numpy().tobytes()).hexdigest()[:12]
print(f"{a.op} {W}x{H}, {a.rounds} interleaved rounds x {a.reps} launches")
for n, _ in variants:
    t = times[n]; med = statistics.median(t)
    print(f"{n:28s} med {med:.4f} ms  min {min(t):.4f}  max {max(t):.4f}  {bpp*W*H/med/1e6:7.0f} GB/s  {bpp*W*H/med/8e7:5.1f}%  sha {shas[n]}")
