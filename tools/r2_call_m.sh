#!/bin/bash
# round 2, GPU call M: band height / residency / order for the shard sizes of 2, 4 and 8 GPUs (8192, 4096, 2048 rows of a 16384-wide frame)
set -o pipefail
O=gpurun_out/r2m
mkdir -p $O
for h in 2048 4096 8192; do
  V=()
  for r in 16 32 64; do for o in 0 6; do for pad in 0 30; do V+=("R${r}_o${o}_p${pad}=GH_FUSED_DMA_R:${r},GH_FUSED_DMA_ORDER:${o},GH_FUSED_DMA_PAD_KB:${pad}"); done; done; done
  V+=("R16_d3=GH_FUSED_DMA_R:16,GH_FUSED_DMA:3" "R32_d3=GH_FUSED_DMA_R:32,GH_FUSED_DMA:3" "R8_o0=GH_FUSED_DMA_R:8,GH_FUSED_DMA_ORDER:0" "R24_o0=GH_FUSED_DMA_R:24,GH_FUSED_DMA_ORDER:0")
  echo "== fused 16384 x $h"
  python3 tools/ab_fused.py --op fused --w 16384 --h $h --rounds 5 --reps 40 "${V[@]}" 2>&1 | grep -v amdgpu.ids | sort -k3 -n | head -12
done | tee $O/ab_fused_shards.txt
for h in 2048 4096; do
  V=()
  for r in 16 32 64; do for o in 0 8; do V+=("R${r}_o${o}=GH_SOBEL_R:${r},GH_SOBEL_ORDER:${o}"); done; done
  echo "== sobel 16384 x $h"
  python3 tools/ab_fused.py --op sobel --w 16384 --h $h --rounds 5 --reps 40 "${V[@]}" 2>&1 | grep -v amdgpu.ids | sort -k3 -n | head -8
done | tee $O/ab_sobel_shards.txt
