#!/usr/bin/env python3
"""A few single-frame sharded steps in loopback rehearsal, for `rocprofv3 --kernel-trace`: which kernel runs when.  Developer tool.
    rocprofv3 --kernel-trace -d gpurun_out/prof -o steps --output-format csv -- python3 tools/shard_step_trace.py 2048"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("openmp-parallel-computing_amd"); api = pkg.hostapi; api.init(0)
comm = pkg.sharding.NativeHaloComm(0, 1); comm.set_loopback(True)
rows, W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 16384
buf = torch.randint(0, 256, (rows + 2, W, 3), dtype=torch.uint8, device="cuda")
edge = torch.empty((rows, W), dtype=torch.uint8, device="cuda")
step = comm.bind_fused_step(buf, edge, rows, 0)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    step()
torch.cuda.synchronize()
comm.close()
