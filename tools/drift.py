#!/usr/bin/env python3
"""Per-launch time of one kernel over a long back-to-back run (developer tool): does the GPU hold its speed?"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("openmp-parallel-computing_amd"); api = pkg.hostapi; api.init(0)
op = sys.argv[1] if len(sys.argv) > 1 else "fused"
W = H = 16384
rgb = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda")
plane = torch.randint(0, 256, (H, W), dtype=torch.uint8, device="cuda")
out = torch.empty((H, W), dtype=torch.uint8, device="cuda")
fn = (lambda: api.gray_sobel_fused(rgb, 0, out=out)) if op == "fused" else (lambda: api.sobel(plane, out=out))
chunk, nchunks = 25, int(sys.argv[2]) if len(sys.argv) > 2 else 240
ev = [torch.cuda.Event(enable_timing=True) for _ in range(nchunks + 1)]
torch.cuda.synchronize(); time.sleep(1.0)          # start from an idle GPU
ev[0].record()
for c in range(nchunks):
    for _ in range(chunk): fn()
    ev[c + 1].record()
torch.cuda.synchronize()
ts = [ev[c].elapsed_time(ev[c + 1]) / chunk for c in range(nchunks)]
for c in range(0, nchunks, max(1, nchunks // 24)):
    print(f"launch {c*chunk:6d}  t={sum(ts[:c])*chunk/1e3:6.3f} s   {ts[c]*1e3:7.1f} us/launch")
print(f"first chunk {ts[0]*1e3:.1f} us, best {min(ts)*1e3:.1f}, median {sorted(ts)[len(ts)//2]*1e3:.1f}, last {ts[-1]*1e3:.1f}")
