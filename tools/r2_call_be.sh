#!/bin/bash
# round 2, GPU call BE: band height of the fused kernel on LARGE frames whose pitch is off-step (16368, 15360, 12288, 10240 wide)
set -o pipefail
O=gpurun_out/r2be
mkdir -p $O
for wh in "16368 16384" "15360 16384" "12288 16384" "10240 16384" "16370 16384" "16384 16384"; do
  set -- $wh
  echo "== fused $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op fused --w $1 --h $2 --rounds 5 --in-ring 3 --out-ring 6 "r16=GH_FUSED_DMA_R:16" "r32=GH_FUSED_DMA_R:32" "r16_pad0=GH_FUSED_DMA_R:16,GH_FUSED_DMA_PAD_KB:0" "r32_pad0=GH_FUSED_DMA_R:32,GH_FUSED_DMA_PAD_KB:0" "r16_d3=GH_FUSED_DMA_R:16,GH_FUSED_DMA:3" 2>&1 | grep -v amdgpu.ids | tail -6
done | tee $O/fused_r_large.txt
