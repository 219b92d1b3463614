#!/bin/bash
# round 2, GPU call AP: fused kernel on shard sizes with ROTATING buffers (a 4096-row shard fits the Infinity Cache otherwise): band height x ring depth
set -o pipefail
O=gpurun_out/r2ap
mkdir -p $O
for h in 4096 2048 8192; do
  V=()
  for r in 16 32 64; do for d in 2 3 4; do V+=("R${r}_d${d}=GH_FUSED_DMA_R:${r},GH_FUSED_DMA:${d}"); done; done
  V+=("R32_d3_p30=GH_FUSED_DMA_R:32,GH_FUSED_DMA:3,GH_FUSED_DMA_PAD_KB:30" "R16_d3_p30=GH_FUSED_DMA_R:16,GH_FUSED_DMA:3,GH_FUSED_DMA_PAD_KB:30" "R16_d2_p30=GH_FUSED_DMA_R:16,GH_FUSED_DMA:2,GH_FUSED_DMA_PAD_KB:30")
  echo "== fused 16384 x $h"
  timeout -k 10 300 python3 tools/ab_fused.py --op fused --w 16384 --h $h --rounds 5 --reps 30 --in-ring 4 --out-ring 6 "${V[@]}" 2>&1 | grep -v amdgpu.ids | sort -k3 -n | head -14
done | tee $O/ab_fused_shards_ring.txt
