#!/bin/bash
# round 2, GPU call AQ: regular-band (branch-free) form of the fused kernel: full frame and shard sizes, rotating buffers
set -o pipefail
O=gpurun_out/r2aq
mkdir -p $O
for h in 16384 8192 4096 2048; do
  V=("generic=GH_FUSED_REG:0")
  if [ $h -le 4096 ]; then
    V+=("R16_d3=GH_FUSED_DMA_R:16,GH_FUSED_DMA:3" "R16_d3_generic=GH_FUSED_DMA_R:16,GH_FUSED_DMA:3,GH_FUSED_REG:0" "R64_d3=GH_FUSED_DMA_R:64,GH_FUSED_DMA:3" "R32_d2=GH_FUSED_DMA_R:32,GH_FUSED_DMA:2" "R32_d2_generic=GH_FUSED_DMA_R:32,GH_FUSED_DMA:2,GH_FUSED_REG:0" "R32_d3_generic=GH_FUSED_DMA_R:32,GH_FUSED_DMA:3")
  else
    V+=("d3=GH_FUSED_DMA:3" "d3_generic=GH_FUSED_DMA:3,GH_FUSED_REG:0" "d3_pad0=GH_FUSED_DMA:3,GH_FUSED_DMA_PAD_KB:0" "d3_pad30=GH_FUSED_DMA:3,GH_FUSED_DMA_PAD_KB:30")
  fi
  echo "== fused 16384 x $h"
  timeout -k 10 300 python3 tools/ab_fused.py --op fused --w 16384 --h $h --rounds 7 --reps 20 --in-ring 4 --out-ring 6 "${V[@]}" 2>&1 | grep -v amdgpu.ids | tail -8
done | tee $O/ab_fused_reg.txt
