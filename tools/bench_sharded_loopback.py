#!/usr/bin/env python3
"""Host and GPU cost of one natively issued sharded step (csrc/comm.hip) for the shard sizes of a 16384^2 frame split
over 1..8 GPUs, rehearsed on one GPU with the loopback communicator (ncclSend/ncclRecv to self).  Developer tool."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("openmp-parallel-computing_amd"); api = pkg.hostapi; api.init(0)
comm = pkg.sharding.NativeHaloComm(0, 1)
LB = int(os.environ.get("GH_LOOPBACK_MODE", "1"))      # 1 middle shard, 2 top shard (frame border above, one halo row), 3 bottom shard
comm.set_loopback(LB)
if os.environ.get("GH_STEP_FORM"):                      # the form through the API (the release library reads GH_SHARD_STEP_FORM once per process)
    comm.set_step_form(int(os.environ["GH_STEP_FORM"]))
KIND = {1: "middle", 2: "top", 3: "bottom"}[LB]
W = 16384
WORLDS = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4, 8]
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""
NSTEP = int(sys.argv[3]) if len(sys.argv) > 3 else 300
for world in WORLDS:
    rows = 16384 // world
    buf = torch.randint(0, 256, (rows + 2, W, 3), dtype=torch.uint8, device="cuda")
    edge = torch.empty((rows, W), dtype=torch.uint8, device="cuda")
    step = comm.bind_fused_step(buf, edge, rows, 0)
    whole = lambda: api.gray_sobel_fused_rows(buf, edge, rows, 0, LB == 2, LB == 3)      # the shard kernel alone: halo rows in place, border where the shard has one
    buf2, buf3 = buf.clone(), buf.clone()
    pstep = comm.bind_fused_step_pipelined([buf, buf2], edge, rows, 0)
    pstep3 = comm.bind_fused_step_pipelined([buf, buf2, buf3], edge, rows, 0)
    ring8 = [buf, buf2, buf3] + [buf.clone() for _ in range(5)]
    bstep4 = comm.bind_fused_step_batched(ring8, edge, rows, 0, 4)
    bstep2 = comm.bind_fused_step_batched(ring8, edge, rows, 0, 2)
    sstep8 = comm.bind_fused_stream_step(ring8, edge, rows, 0)
    def stream_flag8():
        sstep8(cnt[0]); cnt[0] += 1
    ring_steps = [comm.bind_fused_step(b, edge, rows, 0) for b in ring8]
    def stream8():
        ring_steps[cnt[0] % 8](); cnt[0] += 1
    cnt = [0]
    def batched4():
        bstep4(cnt[0]); cnt[0] += 1
    def batched2():
        bstep2(cnt[0]); cnt[0] += 1
    def piped():
        pstep(cnt[0]); cnt[0] += 1
    def piped3():
        pstep3(cnt[0]); cnt[0] += 1
    for name, fn in (("native sharded step (GH_SHARD_STEP_FORM: 2 halo flag, 1 serial, 0 events)", step), ("native pipelined step, halo 1 frame ahead", piped),
                     ("native pipelined step, halo 2 frames ahead", piped3), ("native batched step, halos of 4 frames per exchange (ring of 8)", batched4),
                     ("native batched step, halos of 2 frames per exchange (ring of 8)", batched2), ("single-frame step over a ring of 8 different frames", stream8), ("stream step (halo of the next frame beside this launch, flags; ring of 8)", stream_flag8), ("single launch, no exchange", whole)):
        if ONLY and ONLY not in name: continue
        cnt[0] = 0
        for _ in range(24): fn()
        torch.cuda.synchronize()
        n = NSTEP
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); a.record()
        for _ in range(n): fn()
        t1 = time.perf_counter(); b.record(); b.synchronize(); t2 = time.perf_counter()
        print(f"{KIND:6s} shard of 1/{world} ({rows} rows): {name:62s} host enqueue {1e6*(t1-t0)/n:7.1f} us/step  wall {1e6*(t2-t0)/n:7.1f} us/step  "
              f"gpu {1e3*a.elapsed_time(b)/n:7.1f} us/step  -> {W*rows/( (t2-t0)/n)/1e6:9.0f} Mpx/s per rank")
comm.close()
