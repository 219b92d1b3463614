#!/bin/bash
# round 2, GPU call AY: regular one-strip Sobel kernel vs the generic one at EQUAL band height, and vs two strips per wave, on shards and mid-size frames
set -o pipefail
O=gpurun_out/r2ay
mkdir -p $O
export GRAYHIP_LIB=$PWD/build_probes/libgrayhip_probes.so
for wh in "12288 8192" "16384 2048" "16384 4096" "16384 8192" "8192 8192" "3840 2160" "1920 1088" "32768 8192"; do
  set -- $wh
  echo "== sobel $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op sobel --w $1 --h $2 --rounds 7 --in-ring 6 --out-ring 6 "gen_r16=GH_SOBEL_NS:1,GH_SOBEL_R:16" "reg_r16=GH_SOBEL_NS:1,GH_SOBEL_DMA:63,GH_SOBEL_R:16" "gen_r64=GH_SOBEL_NS:1,GH_SOBEL_R:64" "reg_r64=GH_SOBEL_NS:1,GH_SOBEL_DMA:63,GH_SOBEL_R:64" "two_r16=GH_SOBEL_NS:2,GH_SOBEL_R:16" "two_r64=GH_SOBEL_NS:2,GH_SOBEL_R:64" 2>&1 | grep -v amdgpu.ids | tail -7
done | tee $O/sobel_reg2.txt
