#!/bin/bash
# Grid search over the workgroup -> (band, column) mapping and band height of the LDS-DMA kernels (developer tool, GPU box).
#   tools/search_mapping.sh fused|sobel "<R list>" "<M list>" "<C list>" "<S list>"
OP=${1:-fused}
RS=${2:-"64 128 256"}; MS=${3:-"0 2 4 6"}; CS=${4:-"0 1 2 3"}; SS=${5:-"0 2"}
if [ "$OP" = fused ]; then RK=GH_FUSED_DMA_R; OK=GH_FUSED_DMA_ORDER; else RK=GH_SOBEL_R; OK=GH_SOBEL_ORDER; fi
args=()
for R in $RS; do for M in $MS; do for C in $CS; do for S in $SS; do
  o=$((S*100 + C*10 + M)); args+=("R${R}_o${o}=${RK}:${R},${OK}:${o}")
done; done; done; done
python3 tools/ab_fused.py --op $OP --rounds ${ROUNDS:-3} "${args[@]}" | grep -v interleaved | sort -k3 -n | head -${TOP:-16}
