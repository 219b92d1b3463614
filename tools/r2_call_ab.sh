#!/bin/bash
# round 2, GPU call AB: wide-strip skeletons up to a whole 16 KiB row per wave
set -o pipefail
O=gpurun_out/r2ab
mkdir -p $O
export GRAYHIP_LIB=$PWD/build_probes/libgrayhip_probes.so
timeout -k 10 300 python3 tools/ab_fused.py --op sobel --rounds 7 --in-ring 6 --out-ring 6 "skeleton=GH_SOBEL_DMA:33" \
  "w4_d3=GH_SOBEL_DMA:403" "w8_d3=GH_SOBEL_DMA:803" "w8_d3_r32=GH_SOBEL_DMA:803,GH_SOBEL_R:32" "w8_d3_r16=GH_SOBEL_DMA:803,GH_SOBEL_R:16" "w8_d2_r32=GH_SOBEL_DMA:802,GH_SOBEL_R:32" \
  "w16_d3_r32=GH_SOBEL_DMA:1603,GH_SOBEL_R:32" "w16_d3_r16=GH_SOBEL_DMA:1603,GH_SOBEL_R:16" "w16_d3_r8=GH_SOBEL_DMA:1603,GH_SOBEL_R:8" "w16_d2_r16=GH_SOBEL_DMA:1602,GH_SOBEL_R:16" \
  "w16_d3_r16_o0=GH_SOBEL_DMA:1603,GH_SOBEL_R:16,GH_SOBEL_ORDER:0" "w8_d3_r32_o0=GH_SOBEL_DMA:803,GH_SOBEL_R:32,GH_SOBEL_ORDER:0" > $O/sobel.txt 2>&1 || { tail $O/sobel.txt; exit 1; }
tail -13 $O/sobel.txt
