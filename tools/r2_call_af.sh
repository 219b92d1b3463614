#!/bin/bash
# round 2, GPU call AF: plane Sobel ring depth 2 (136 VGPRs: three workgroups per CU) against depth 3 (174 VGPRs: two)
set -o pipefail
O=gpurun_out/r2af
mkdir -p $O
timeout -k 10 300 python3 tools/ab_fused.py --op sobel --rounds 9 --in-ring 6 --out-ring 6 "d2=GH_SOBEL_DMA:2" "d2_pad40=GH_SOBEL_DMA:2,GH_SOBEL_PAD_KB:40" "d4=GH_SOBEL_DMA:4" "d2_o0=GH_SOBEL_DMA:2,GH_SOBEL_ORDER:0" "d2_o6=GH_SOBEL_DMA:2,GH_SOBEL_ORDER:6" > $O/sobel.txt 2>&1 || { tail $O/sobel.txt; exit 1; }
tail -6 $O/sobel.txt
