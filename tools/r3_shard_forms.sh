#!/bin/bash
# round 3: single-frame sharded step in its three forms (halo flag / serial / events) on the shard sizes of a 16384^2 frame over 8, 4, 2 GPUs,
# loopback rehearsal on one GPU; plus the launch without any exchange
set -o pipefail
out=gpurun_out/r3_shards; mkdir -p $out
for form in 2 1 0; do
  echo "== GH_SHARD_STEP_FORM=$form"
  GH_TUNE_DYNAMIC=1 GH_SHARD_STEP_FORM=$form python3 tools/bench_sharded_loopback.py 8,4,2 "native sharded step" 2>&1 | grep -v amdgpu.ids | grep shard | cut -c1-230
done | tee $out/forms.txt
python3 tools/bench_sharded_loopback.py 8,4,2 "single launch" 2>&1 | grep -v amdgpu.ids | grep shard | cut -c1-230 | tee -a $out/forms.txt
