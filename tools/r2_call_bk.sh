#!/bin/bash
# round 2, GPU call BK: does the aliasing count predict a better band height for the 2048-row shard (16-row bands: 4 phases)?  fused and plane Sobel, odd heights
set -o pipefail
O=gpurun_out/r2bk
mkdir -p $O
echo "== fused 16384 x 2048"
timeout -k 10 300 python3 tools/ab_fused.py --op fused --w 16384 --h 2048 --rounds 7 --in-ring 4 --out-ring 6 "r20=GH_FUSED_DMA_R:20" "r21=GH_FUSED_DMA_R:21" "r22=GH_FUSED_DMA_R:22" "r24=GH_FUSED_DMA_R:24" "r28=GH_FUSED_DMA_R:28" "r12=GH_FUSED_DMA_R:12" "r32=GH_FUSED_DMA_R:32" 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/fused_2048.txt
echo "== fused 16384 x 1024"
timeout -k 10 300 python3 tools/ab_fused.py --op fused --w 16384 --h 1024 --rounds 7 --in-ring 4 --out-ring 6 "r10=GH_FUSED_DMA_R:10" "r11=GH_FUSED_DMA_R:11" "r20=GH_FUSED_DMA_R:20" "r21=GH_FUSED_DMA_R:21" "r22=GH_FUSED_DMA_R:22" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/fused_1024.txt
echo "== sobel 16384 x 2048 (one strip, generic)"
timeout -k 10 300 python3 tools/ab_fused.py --op sobel --w 16384 --h 2048 --rounds 7 --in-ring 6 --out-ring 6 "one_r16=GH_SOBEL_NS:1" "one_r20=GH_SOBEL_NS:1,GH_SOBEL_R:20" "one_r21=GH_SOBEL_NS:1,GH_SOBEL_R:21" "one_r22=GH_SOBEL_NS:1,GH_SOBEL_R:22" "one_r28=GH_SOBEL_NS:1,GH_SOBEL_R:28" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/sobel_2048.txt
