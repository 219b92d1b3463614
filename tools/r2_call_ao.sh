#!/bin/bash
# round 2, GPU call AO: would a two-strip fused kernel help on shards?  3:1 skeletons (3 / 6 KiB read per wave and row) at the shard sizes of 2 / 4 / 8 GPUs
set -o pipefail
O=gpurun_out/r2ao
mkdir -p $O
export GRAYHIP_LIB=$PWD/build_probes/libgrayhip_probes.so
for h in 8192 4096 2048; do
  timeout -k 10 300 python3 tools/ab_fused.py --op fused --h $h --rounds 5 --in-ring 3 --out-ring 6 "d3=GH_FUSED_DMA:3" "skeleton_d3=GH_FUSED_DMA:3,GH_FUSED_DMA_ST:12" > $O/fused_$h.txt 2>&1 || { tail $O/fused_$h.txt; exit 1; }
  echo "== fused rows $h"; tail -3 $O/fused_$h.txt
  timeout -k 10 300 python3 tools/ab_fused.py --op sobel --w 49152 --h $h --rounds 5 --in-ring 3 --out-ring 3 \
    "p3_d2=GH_SOBEL_DMA:30302" "p3_d3=GH_SOBEL_DMA:30303" "p6_d2=GH_SOBEL_DMA:30602" "p6_d3=GH_SOBEL_DMA:30603" "p6_d2_r16=GH_SOBEL_DMA:30602,GH_SOBEL_R:16" "p3_d2_r16=GH_SOBEL_DMA:30302,GH_SOBEL_R:16" "p6_d3_r16=GH_SOBEL_DMA:30603,GH_SOBEL_R:16" > $O/wide31_$h.txt 2>&1 || { tail $O/wide31_$h.txt; exit 1; }
  echo "== 3:1 skeletons rows $h"; tail -7 $O/wide31_$h.txt
done
