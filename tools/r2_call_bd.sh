#!/bin/bash
# round 2, GPU call BD: band height of the fused kernel on mid-size frames (its rule was derived on 16384-wide frames and shards)
set -o pipefail
O=gpurun_out/r2bd
mkdir -p $O
for wh in "12288 4096" "7680 4352" "12288 8192" "4096 8192" "8192 8192" "5000 3008" "10240 10240"; do
  set -- $wh
  echo "== fused $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op fused --w $1 --h $2 --rounds 7 --in-ring 4 --out-ring 6 "r16=GH_FUSED_DMA_R:16" "r32=GH_FUSED_DMA_R:32" "r64=GH_FUSED_DMA_R:64" "r16_d2=GH_FUSED_DMA_R:16,GH_FUSED_DMA:2" "r64_d3=GH_FUSED_DMA_R:64,GH_FUSED_DMA:3" 2>&1 | grep -v amdgpu.ids | tail -6
done | tee $O/fused_r.txt
