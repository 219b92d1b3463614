#!/bin/bash
# round 2, GPU call AX: regular-band (branch-free) form of the ONE-strip Sobel kernel (134 / 128 VGPRs instead of 174): off-step pitches and small frames
set -o pipefail
O=gpurun_out/r2ax
mkdir -p $O
export GRAYHIP_LIB=$PWD/build_probes/libgrayhip_probes.so
for wh in "16370 16384" "16368 16384" "16384 16384" "4096 4096" "16384 1024" "2048 2048" "7680 4352" "12288 8192"; do
  set -- $wh
  echo "== sobel $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op sobel --w $1 --h $2 --rounds 7 --in-ring 6 --out-ring 6 "one_strip=GH_SOBEL_NS:1" "one_strip_reg=GH_SOBEL_NS:1,GH_SOBEL_DMA:63" "one_strip_reg_r16=GH_SOBEL_NS:1,GH_SOBEL_DMA:63,GH_SOBEL_R:16" 2>&1 | grep -v amdgpu.ids | tail -4
done | tee $O/sobel_reg1.txt
