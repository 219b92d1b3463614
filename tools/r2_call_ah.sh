#!/bin/bash
# round 2, GPU call AH: two strips per wave, deeper rings capped at 256 VGPRs (spills outside the steady loop) against depth 2
set -o pipefail
O=gpurun_out/r2ah
mkdir -p $O
timeout -k 10 300 python3 tools/ab_fused.py --op sobel --rounds 7 --in-ring 6 --out-ring 6 "ns2_d2=GH_SOBEL_NS:2,GH_SOBEL_NS_D:2" "ns2_d3cap=GH_SOBEL_NS:2,GH_SOBEL_NS_D:32" "ns2_d4cap=GH_SOBEL_NS:2,GH_SOBEL_NS_D:42" \
  "ns2_d3cap_o6=GH_SOBEL_NS:2,GH_SOBEL_NS_D:32,GH_SOBEL_ORDER:6" "ns2_d3cap_o0=GH_SOBEL_NS:2,GH_SOBEL_NS_D:32,GH_SOBEL_ORDER:0" > $O/sobel.txt 2>&1 || { tail $O/sobel.txt; exit 1; }
tail -7 $O/sobel.txt
