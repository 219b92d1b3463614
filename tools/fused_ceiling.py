#!/usr/bin/env python3
"""The headline kernel beside its neighbours (round 4): k_fused_dma as shipped (ring depth 2 + residency pad), at depth 3, its no-arithmetic
skeleton, the LINEAR kernel that moves the same bytes (RGB -> luma plane, 3 R + 1 W) and a linear copy -- 16384^2 RGB8, outputs rotating over
three planes, HIP-event timed; under `rocprofv3 --pmc` (tools/profile_sobel_ceiling.sh with GH_CEILING_TOOL=tools/fused_ceiling.py) every
variant's counters in one pass.  Probe build."""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GH_TUNE_DYNAMIC"] = "1"
os.environ.setdefault("GRAYHIP_LIB", os.path.join(ROOT, "build_probes", "libgrayhip_probes.so"))
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=120)
ap.add_argument("--rounds", type=int, default=1)
a = ap.parse_args()
pkg = importlib.import_module("openmp-parallel-computing_amd"); api = pkg.hostapi; api.init(0)
lib = pkg.abi.load()
W = H = 16384
torch.manual_seed(5)
rgb = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda")
rgb2 = torch.empty_like(rgb)
outs = [torch.empty((H, W), dtype=torch.uint8, device="cuda") for _ in range(3)]
s = torch.cuda.current_stream().cuda_stream
KNOBS = ["GH_FUSED_DMA", "GH_FUSED_DMA_ST", "GH_FUSED_DMA_PAD_KB"]
VARIANTS = [("fused, ring depth 2 + pad (ships)", {}, "fused", 4), ("fused, ring depth 3, no pad", {"GH_FUSED_DMA": "3", "GH_FUSED_DMA_PAD_KB": "0"}, "fused", 4),
            ("fused, no arithmetic (skeleton, depth 3)", {"GH_FUSED_DMA": "3", "GH_FUSED_DMA_ST": "12"}, "fused", 4),
            ("linear RGB -> luma plane (3 R + 1 W)", {}, "gray", 4), ("linear copy of the RGB frame (3 R + 3 W)", {}, "copy", 6)]
res = {}
for rnd in range(a.rounds):
    for name, env, op, bpp in VARIANTS:
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)
        def launch(i):
            if op == "fused":
                api.gray_sobel_fused(rgb, 0, out=outs[i % 3])
            elif op == "gray":
                api.gray_to_plane(rgb, 0, out=outs[i % 3])
            else:
                pkg.abi.check(lib.gh_copy_stream_dev(rgb.data_ptr(), rgb2.data_ptr(), rgb.numel(), s), "copy")
        for i in range(40):
            launch(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.launches):
            launch(i)
        e1.record(); e1.synchronize()
        res.setdefault(name, []).append((e0.elapsed_time(e1) / a.launches, bpp, api.last_kernel() if op == "fused" else op))
print(f"{'variant':44s} {'kernel form':28s} {'ms (per round)':>28s} {'% of 8 TB/s':>12s}")
for name, v in res.items():
    ms = [x[0] for x in v]
    print(f"{name:44s} {v[0][2]:28s} {' '.join(f'{m:.4f}' for m in ms):>28s} {v[0][1] * W * H / min(ms) / 1e6 / 80:12.1f}")
