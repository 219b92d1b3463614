#!/bin/bash
# round 2, GPU call AA: wide-strip skeletons (a wave owns 2 or 4 KiB of every row) against the 1 KiB skeleton and the real kernel
set -o pipefail
O=gpurun_out/r2aa
mkdir -p $O
export GRAYHIP_LIB=$PWD/build_probes/libgrayhip_probes.so
timeout -k 10 300 python3 tools/ab_fused.py --op sobel --rounds 7 --in-ring 6 --out-ring 6 "skeleton=GH_SOBEL_DMA:33" \
  "w2_d3=GH_SOBEL_DMA:203" "w2_d2=GH_SOBEL_DMA:202" "w4_d3=GH_SOBEL_DMA:403" "w4_d2=GH_SOBEL_DMA:402" \
  "w2_d3_r32=GH_SOBEL_DMA:203,GH_SOBEL_R:32" "w4_d3_r32=GH_SOBEL_DMA:403,GH_SOBEL_R:32" "w4_d3_r16=GH_SOBEL_DMA:403,GH_SOBEL_R:16" \
  "w2_d3_o0=GH_SOBEL_DMA:203,GH_SOBEL_ORDER:0" "w4_d3_o0=GH_SOBEL_DMA:403,GH_SOBEL_ORDER:0" "w2_d3_o6=GH_SOBEL_DMA:203,GH_SOBEL_ORDER:6" > $O/sobel.txt 2>&1 || { tail $O/sobel.txt; exit 1; }
tail -13 $O/sobel.txt
