#!/bin/bash
# round 2, GPU call B: is the default-policy gain of the plane Sobel an Infinity-Cache effect?  (rotating inputs / outputs)
set -o pipefail
O=gpurun_out/r2b
mkdir -p $O
export GRAYHIP_LIB=$PWD/build_probes/libgrayhip_probes.so
V=("lp0=GH_SOBEL_DMA:23" "R16_xcd_lp0=GH_SOBEL_R:16,GH_SOBEL_ORDER:4,GH_SOBEL_DMA:23" "R32_xcd_lp0=GH_SOBEL_R:32,GH_SOBEL_ORDER:4,GH_SOBEL_DMA:23" \
   "R32_lp0=GH_SOBEL_R:32,GH_SOBEL_DMA:23" "R128_lp0=GH_SOBEL_R:128,GH_SOBEL_DMA:23" "R128=GH_SOBEL_R:128" "lp0_o0=GH_SOBEL_DMA:23,GH_SOBEL_ORDER:0" "lp0_o6=GH_SOBEL_DMA:23,GH_SOBEL_ORDER:6" \
   "lp0_d2=GH_SOBEL_DMA:22" "skeleton=GH_SOBEL_DMA:33")
python3 tools/ab_fused.py --op sobel --rounds 5 "${V[@]}" > $O/ab_sobel_ring1.txt 2>&1; echo "ring1 rc=$?"
python3 tools/ab_fused.py --op sobel --rounds 5 --in-ring 6 --out-ring 6 "${V[@]}" > $O/ab_sobel_ring6.txt 2>&1; echo "ring6 rc=$?"
python3 tools/ab_fused.py --op sobel --rounds 5 --in-ring 6 --out-ring 1 "${V[@]}" > $O/ab_sobel_in6.txt 2>&1; echo "in6 rc=$?"
python3 tools/ab_fused.py --op gray --rounds 5 --in-ring 3 --out-ring 6 > $O/ab_gray_ring.txt 2>&1
python3 tools/ab_fused.py --op fused --rounds 5 --in-ring 3 --out-ring 6 "lp0=GH_FUSED_DMA:2,GH_FUSED_DMA_ST:102,GH_FUSED_DMA_PAD_KB:30" > $O/ab_fused_ring.txt 2>&1
cat $O/ab_sobel_ring1.txt $O/ab_sobel_ring6.txt $O/ab_sobel_in6.txt $O/ab_gray_ring.txt $O/ab_fused_ring.txt
