#!/bin/bash
# round 2, GPU call AI: two strips per wave, regular-band form at ring depth 3 (202 VGPRs) against depth 2 (223) and the one-strip kernel
set -o pipefail
O=gpurun_out/r2ai
mkdir -p $O
timeout -k 10 300 python3 tools/ab_fused.py --op sobel --rounds 9 --in-ring 6 --out-ring 6 "ns2_d2=GH_SOBEL_NS:2,GH_SOBEL_NS_D:2" "ns2_reg=GH_SOBEL_NS:2,GH_SOBEL_NS_D:33" \
  "ns2_reg_o6=GH_SOBEL_NS:2,GH_SOBEL_NS_D:33,GH_SOBEL_ORDER:6" "ns2_reg_o0=GH_SOBEL_NS:2,GH_SOBEL_NS_D:33,GH_SOBEL_ORDER:0" "ns2_reg_r16=GH_SOBEL_NS:2,GH_SOBEL_NS_D:33,GH_SOBEL_R:16" > $O/sobel.txt 2>&1 || { tail $O/sobel.txt; exit 1; }
tail -7 $O/sobel.txt
for h in 8192 2048; do
timeout -k 10 300 python3 tools/ab_fused.py --op sobel --w 16384 --h $h --rounds 7 --in-ring 6 --out-ring 6 "ns2_d2=GH_SOBEL_NS:2,GH_SOBEL_NS_D:2" "ns2_reg=GH_SOBEL_NS:2,GH_SOBEL_NS_D:33" "ns2_reg_r16=GH_SOBEL_NS:2,GH_SOBEL_NS_D:33,GH_SOBEL_R:16" "base_r16=GH_SOBEL_R:16" > $O/sobel_$h.txt 2>&1 || { tail $O/sobel_$h.txt; exit 1; }
echo "== rows $h"; tail -5 $O/sobel_$h.txt
done
