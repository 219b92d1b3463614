#!/bin/bash
# round 2, GPU call AR: fused kernel, 32-row bands as soon as they give every CU a workgroup? (no: 16-row bands win on every shape the change would touch -- rule left as it was)
set -o pipefail
O=gpurun_out/r2ar
mkdir -p $O
for wh in "16384 2048" "8192 4096" "4096 8192" "12288 2048" "16384 3072"; do
  set -- $wh
  echo "== fused $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op fused --w $1 --h $2 --rounds 7 --reps 20 --in-ring 4 --out-ring 6 "R16=GH_FUSED_DMA_R:16" "R32=GH_FUSED_DMA_R:32" 2>&1 | grep -v amdgpu.ids | tail -3
done | tee $O/ab_fused_r32.txt
