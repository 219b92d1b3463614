#!/bin/bash
# round 2, GPU call BC: band height of the convolution's ring kernel (it had inherited the Sobel rule untested): 16 / 32 / 64 rows, regular form where it applies
set -o pipefail
O=gpurun_out/r2bc
mkdir -p $O
for wh in "16384 16384" "32768 8192" "16384 8192" "8192 8192" "16384 4096" "16370 16384" "4096 16384"; do
  set -- $wh
  echo "== conv $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op conv --w $1 --h $2 --rounds 7 "r16=GH_CONV_R:16" "r32=GH_CONV_R:32" "r64=GH_CONV_R:64" "r16_gen=GH_CONV_R:16,GH_CONV_REG:0" "r16_o0=GH_CONV_R:16,GH_CONV_ORDER:0" "r16_o6=GH_CONV_R:16,GH_CONV_ORDER:6" 2>&1 | grep -v amdgpu.ids | tail -7
done | tee $O/conv_r.txt
