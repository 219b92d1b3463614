#!/bin/bash
# round 2, GPU call AV: batched halo prefetch (one exchange per 2 / 4 frames): loopback parity tests, then step costs at the shard sizes of 2 / 4 / 8 GPUs
set -o pipefail
O=gpurun_out/r2av
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "batched_halo or pipelined_steps_loopback or sharded_steps_loopback" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 400 python3 tools/bench_sharded_loopback.py 2,4,8 "" 400 > $O/loopback.txt 2>&1 || { tail $O/loopback.txt; exit 1; }
grep -v amdgpu $O/loopback.txt | cut -c1-200
