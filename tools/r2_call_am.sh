#!/bin/bash
# round 2, GPU call AM: two-strip regular-band Sobel kernel, ring depths 2 / 3 / 6 and dispatch orders (probes build)
set -o pipefail
O=gpurun_out/r2am
mkdir -p $O
export GRAYHIP_LIB=$PWD/build_probes/libgrayhip_probes.so
timeout -k 10 300 python3 tools/ab_fused.py --op sobel --rounds 9 --in-ring 6 --out-ring 6 "one_strip=GH_SOBEL_NS:1" "reg_d2=GH_SOBEL_NS_D:22" "reg_d6=GH_SOBEL_NS_D:66" \
  "reg_d6_pad=GH_SOBEL_NS_D:66,GH_SOBEL_PAD_KB:32" "reg_d3_o6=GH_SOBEL_ORDER:6" "reg_d3_o0=GH_SOBEL_ORDER:0" "reg_d3_o2=GH_SOBEL_ORDER:2" "reg_d3_o4=GH_SOBEL_ORDER:4" > $O/sobel.txt 2>&1 || { tail $O/sobel.txt; exit 1; }
tail -10 $O/sobel.txt
