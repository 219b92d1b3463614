#!/usr/bin/env python3
"""The plane Sobel's ceiling, one bounded attempt (round 4): the shipped kernel (two strips per wave, ring depth 3,
regular 64-row bands) beside the variants that put more bytes in flight per CU (ring depth 6) or fewer (depth 2), the one-strip kernel
(3 waves per SIMD in its regular form), its no-arithmetic skeleton, and a linear copy of the same two planes -- each `--launches` times
on a 16384^2 plane, inputs and outputs rotating over three planes, HIP-event timed.  Run under `rocprofv3 --pmc ...` the same process
gives every variant's counters in one pass (the variants are different kernel symbols); tools/profile_sobel_ceiling.sh drives the passes.
Probe build (`make probes`): the depth variants and the skeleton are not in the release library."""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GH_TUNE_DYNAMIC"] = "1"
os.environ.setdefault("GRAYHIP_LIB", os.path.join(ROOT, "build_probes", "libgrayhip_probes.so"))
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=150)
ap.add_argument("--size", type=int, default=16384)
ap.add_argument("--rounds", type=int, default=1, help="repeat the whole list this many times (interleaved A/B)")
a = ap.parse_args()
pkg = importlib.import_module("openmp-parallel-computing_amd"); api = pkg.hostapi; api.init(0)
lib = pkg.abi.load()
W = H = a.size
torch.manual_seed(3)
planes = [torch.randint(0, 256, (H, W), dtype=torch.uint8, device="cuda") for _ in range(3)]
outs = [torch.empty((H, W), dtype=torch.uint8, device="cuda") for _ in range(3)]
s = torch.cuda.current_stream().cuda_stream
KNOBS = ["GH_SOBEL_NS_D", "GH_SOBEL_NS", "GH_SOBEL_DMA", "GH_SOBEL_R", "GH_SOBEL_PAD_KB"]
VARIANTS = [("dma2 depth 3 (ships)", {}, "sobel"), ("dma2 depth 6", {"GH_SOBEL_NS_D": "66"}, "sobel"), ("dma2 depth 2", {"GH_SOBEL_NS_D": "22"}, "sobel"),
            ("one strip per wave, generic R=64", {"GH_SOBEL_NS": "1"}, "sobel"), ("one strip, regular R=16 (3 waves/SIMD)", {"GH_SOBEL_NS": "1", "GH_SOBEL_R": "16"}, "sobel"),
            ("one strip, no arithmetic (skeleton)", {"GH_SOBEL_NS": "1", "GH_SOBEL_DMA": "33"}, "sobel"),
            ("skeleton without the halo dword requests", {"GH_SOBEL_NS": "1", "GH_SOBEL_DMA": "63"}, "sobel"),
            ("pass k: luma on load (ships)", {}, "sog"), ("linear copy of the two planes", {}, "copy")]
res = {}
for rnd in range(a.rounds):
    for name, env, op in VARIANTS:
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)
        def launch(i):
            if op == "sobel":
                api.sobel(planes[i % 3], out=outs[i % 3])
            elif op == "sog":
                api.sobel_of_gray(planes[i % 3], 0, out=outs[i % 3])
            else:
                pkg.abi.check(lib.gh_copy_stream_dev(planes[i % 3].data_ptr(), outs[i % 3].data_ptr(), W * H, s), "copy")
        for i in range(60):
            launch(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.launches):
            launch(i)
        e1.record(); e1.synchronize()
        res.setdefault(name, []).append((e0.elapsed_time(e1) / a.launches, api.last_kernel() if op != "copy" else "k_copy_stream"))
print(f"{'variant':44s} {'kernel form':40s} {'ms (per round)':>28s} {'% of 8 TB/s':>12s}")
for name, v in res.items():
    ms = [x[0] for x in v]
    best = min(ms)
    print(f"{name:44s} {v[0][1]:40s} {' '.join(f'{m:.4f}' for m in ms):>28s} {2 * W * H / best / 1e6 / 80:12.1f}")
