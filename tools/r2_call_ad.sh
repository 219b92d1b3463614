#!/bin/bash
# round 2, GPU call AD: 3:1 wide-strip skeletons (the fused kernel's byte shape: 3 / 6 / 12 KiB read and 1 / 2 / 4 KiB written per wave and row)
set -o pipefail
O=gpurun_out/r2ad
mkdir -p $O
export GRAYHIP_LIB=$PWD/build_probes/libgrayhip_probes.so
timeout -k 10 300 python3 tools/ab_fused.py --op fused --rounds 5 --in-ring 3 --out-ring 6 "d3=GH_FUSED_DMA:3" "skeleton_d3=GH_FUSED_DMA:3,GH_FUSED_DMA_ST:12" > $O/fused.txt 2>&1 || { tail $O/fused.txt; exit 1; }
tail -3 $O/fused.txt
timeout -k 10 300 python3 tools/ab_fused.py --op sobel --w 49152 --h 16384 --rounds 5 --in-ring 3 --out-ring 3 \
  "p3_d2=GH_SOBEL_DMA:30302" "p3_d3=GH_SOBEL_DMA:30303" "p6_d2=GH_SOBEL_DMA:30602" "p6_d3=GH_SOBEL_DMA:30603" "p12_d2=GH_SOBEL_DMA:31202" "p12_d3=GH_SOBEL_DMA:31203" \
  "p3_d2_o0=GH_SOBEL_DMA:30302,GH_SOBEL_ORDER:0" "p6_d2_o0=GH_SOBEL_DMA:30602,GH_SOBEL_ORDER:0" "p6_d2_o8=GH_SOBEL_DMA:30602,GH_SOBEL_ORDER:8" > $O/wide31.txt 2>&1 || { tail $O/wide31.txt; exit 1; }
tail -10 $O/wide31.txt
