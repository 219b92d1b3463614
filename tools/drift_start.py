import importlib, os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else ".")
import torch
pkg = importlib.import_module("openmp-parallel-computing_amd"); api = pkg.hostapi; api.init(0)
W = H = 16384
rgb = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda")
out = torch.empty((H, W), dtype=torch.uint8, device="cuda")
fn = lambda: api.gray_sobel_fused(rgb, 0, out=out)
fn(); torch.cuda.synchronize()
for idle in (0.0, 0.05, 0.5, 3.0):
    time.sleep(idle)
    n = 120
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for c in range(n):
        fn(); ev[c + 1].record()
    torch.cuda.synchronize()
    ts = [ev[c].elapsed_time(ev[c + 1]) * 1e3 for c in range(n)]
    print(f"after {idle:4.2f} s idle: launches 0-4: {[round(t) for t in ts[:5]]}  10-14: {[round(t) for t in ts[10:15]]}  20-24: {[round(t) for t in ts[20:25]]}  50-54: {[round(t) for t in ts[50:55]]}  100-104: {[round(t) for t in ts[100:105]]}")
