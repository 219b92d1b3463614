#!/bin/bash
# round 2, GPU call N: does a smaller RCCL point-to-point footprint help the sharded steps? (loopback rehearsal, 8-GPU shard size)
set -o pipefail
O=gpurun_out/r2n
mkdir -p $O
echo "== defaults"; python3 tools/bench_sharded_loopback.py 8,4 2>&1 | grep -v amdgpu.ids | grep shard
echo "== NCCL_MAX_P2P_NCHANNELS=1 NCCL_MIN_P2P_NCHANNELS=1"; NCCL_MAX_P2P_NCHANNELS=1 NCCL_MIN_P2P_NCHANNELS=1 python3 tools/bench_sharded_loopback.py 8,4 2>&1 | grep -v amdgpu.ids | grep shard
echo "== NCCL_MAX_NCHANNELS=2 NCCL_MIN_NCHANNELS=1 + P2P 1"; NCCL_MAX_NCHANNELS=2 NCCL_MIN_NCHANNELS=1 NCCL_MAX_P2P_NCHANNELS=1 NCCL_MIN_P2P_NCHANNELS=1 python3 tools/bench_sharded_loopback.py 8,4 2>&1 | grep -v amdgpu.ids | grep shard
echo "== NCCL_P2P_NET_CHUNKSIZE / NCCL_NTHREADS=64"; NCCL_NTHREADS=64 NCCL_MAX_P2P_NCHANNELS=1 NCCL_MIN_P2P_NCHANNELS=1 python3 tools/bench_sharded_loopback.py 8 2>&1 | grep -v amdgpu.ids | grep shard
