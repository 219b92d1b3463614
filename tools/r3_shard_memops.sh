#!/bin/bash
# round 3: halo-flag form of the single-frame sharded step -- the two side-stream operations as one-wave kernels (0) or as
# command-processor memory operations (1: flag write, 2: started wait, 3: both); loopback rehearsal, shards of 2048 / 4096 rows
set -o pipefail
out=gpurun_out/r3_shards; mkdir -p $out
for m in 0 1 2 3 0 3; do
  echo "== GH_HALO_MEMOPS=$m"
  GH_TUNE_DYNAMIC=1 GH_SHARD_STEP_FORM=2 GH_HALO_MEMOPS=$m timeout -k 5 100 python3 tools/bench_sharded_loopback.py 8,4 "native sharded step" 2>&1 | grep -v amdgpu.ids | grep -E "shard|rror" | cut -c1-230
done | tee $out/memops.txt
