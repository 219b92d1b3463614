#!/usr/bin/env python3
"""Developer probe: which rows land where when the halo rows of several buffers are exchanged in ONE ncclGroup with the
loopback communicator (several sends and receives to the same peer in one group).  Row 1 of buffer i holds 10 + i, its last
row 20 + i; after the exchange halo row 0 should hold 10 + i and halo row rows + 1 should hold 20 + i."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("openmp-parallel-computing_amd"); pkg.hostapi.init(0)
comm = pkg.sharding.NativeHaloComm(0, 1); comm.set_loopback(True)
lib, chk = pkg.abi.load(), pkg.abi.check
rows, rb = 5, 4096
for n in (1, 2, 3, 4, 8):
    bufs = []
    for i in range(n):
        b = torch.zeros((rows + 2, rb), dtype=torch.uint8, device="cuda")
        b[1] = 10 + i
        b[rows] = 20 + i
        b[0] = 99
        b[rows + 1] = 98
        bufs.append(b)
    arr = (C.c_void_p * n)(*[b.data_ptr() for b in bufs])
    chk(lib.gh_sharded_prefetch_halos_dev(comm.handle, arr, n, rb, rows, torch.cuda.current_stream().cuda_stream), "prefetch")
    torch.cuda.synchronize(); comm_side = None
    import time; time.sleep(0.2)
    print(n, [(int(b[0, 0]), int(b[rows + 1, 0])) for b in bufs], flush=True)
comm.close()

# second probe: a batch request for buffers 2, 3 followed by a pipelined step on buffer 0 (never requested: exchanged first)
comm = pkg.sharding.NativeHaloComm(0, 1); comm.set_loopback(True)
rows, w = 5, 2064
bufs = []
for i in range(4):
    b = torch.zeros((rows + 2, w, 3), dtype=torch.uint8, device="cuda")
    b[1] = 10 + i; b[rows] = 20 + i; b[0] = 99; b[rows + 1] = 98
    bufs.append(b)
edge = torch.zeros((rows, w), dtype=torch.uint8, device="cuda")
step = comm.bind_fused_step_batched(bufs, edge, rows, 0, 2)
for k in range(6):
    step(k)
    torch.cuda.synchronize()
    print("after step", k, [(int(b[0, 0, 0]), int(b[rows + 1, 0, 0])) for b in bufs], flush=True)
comm.close()
