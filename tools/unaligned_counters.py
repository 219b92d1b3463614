#!/usr/bin/env python3
"""Launches the fused and the plane-Sobel kernel N times each on a W x H frame (developer tool; run under rocprofv3 --pmc ...,
once per width, to compare the memory-pipeline counters of aligned and unaligned row pitches: tools/sweep.py r2_s)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
W, H, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pkg = importlib.import_module("openmp-parallel-computing_amd"); api = pkg.hostapi; api.init(0)
torch.manual_seed(1)
rgb = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
plane = [torch.randint(0, 256, (H, W), dtype=torch.uint8, device="cuda") for _ in range(3)]
outs = [torch.empty((H, W), dtype=torch.uint8, device="cuda") for _ in range(3)]
for i in range(N):
    api.gray_sobel_fused(rgb[i % 2], 0, out=outs[i % 3])
    api.sobel(plane[i % 3], out=outs[(i + 1) % 3])
torch.cuda.synchronize()
print("done", W, H, N)
