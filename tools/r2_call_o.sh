#!/bin/bash
# round 2, GPU call O: single-frame sharded step -- overlapped (side stream + events + 3 launches) vs serial (exchange, then one launch)
set -o pipefail
mkdir -p gpurun_out/r2o
for mode in 0 1; do
  echo "== GH_SHARD_STEP_SERIAL=$mode"
  GH_TUNE_DYNAMIC=1 GH_SHARD_STEP_SERIAL=$mode python3 tools/bench_sharded_loopback.py 8,4,2 "native sharded step" 2>&1 | grep -v amdgpu.ids | grep shard | cut -c1-220
done | tee gpurun_out/r2o/serial_vs_overlap.txt
python3 tools/bench_sharded_loopback.py 8,4,2 "single launch" 2>&1 | grep -v amdgpu.ids | grep shard | cut -c1-220 | tee -a gpurun_out/r2o/serial_vs_overlap.txt
