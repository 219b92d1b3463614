#!/bin/bash
# round 2, GPU call AG: two strips per wave (k_sobel_dma2), ring depths 2 / 3 / 4 (223 / 326 / 292 VGPRs)
set -o pipefail
O=gpurun_out/r2ag
mkdir -p $O
timeout -k 10 300 python3 tools/ab_fused.py --op sobel --rounds 7 --in-ring 6 --out-ring 6 "ns2_d2=GH_SOBEL_NS:2,GH_SOBEL_NS_D:2" "ns2_d3=GH_SOBEL_NS:2,GH_SOBEL_NS_D:3" "ns2_d4=GH_SOBEL_NS:2,GH_SOBEL_NS_D:4" \
  "ns2_d2_o0=GH_SOBEL_NS:2,GH_SOBEL_NS_D:2,GH_SOBEL_ORDER:0" "ns2_d2_o6=GH_SOBEL_NS:2,GH_SOBEL_NS_D:2,GH_SOBEL_ORDER:6" "ns2_d2_r32=GH_SOBEL_NS:2,GH_SOBEL_NS_D:2,GH_SOBEL_R:32" > $O/sobel.txt 2>&1 || { tail $O/sobel.txt; exit 1; }
tail -8 $O/sobel.txt
