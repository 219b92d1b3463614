#!/bin/bash
# Grid search over the workgroup -> (band, column) mapping and band height of the LDS-DMA kernels (developer tool, GPU box).
OP=${1:-fused}
if [ "$OP" = fused ]; then RK=GH_FUSED_DMA_R; OK=GH_FUSED_DMA_ORDER; else RK=GH_SOBEL_R; OK=GH_SOBEL_ORDER; fi
args=()
for R in 64 128 256; do for M in 0 2 4; do for C in 0 1 2 3; do for S in 0 2; do
  o=$((S*100 + C*10 + M)); args+=("R${R}_o${o}=${RK}:${R},${OK}:${o}")
done; done; done; done
python3 tools/ab_fused.py --op $OP --rounds 3 "${args[@]}" | sort -k3 -n | head -30
