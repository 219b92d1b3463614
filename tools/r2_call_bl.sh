#!/bin/bash
# round 2, GPU call BL: convolution ring kernel, band heights with many phases (20-22 rows) against 16 (4 phases on 16384-wide RGB) and 64 (one)
set -o pipefail
O=gpurun_out/r2bl
mkdir -p $O
for wh in "16384 16384" "12288 12288"; do
  set -- $wh
  echo "== conv $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op conv --w $1 --h $2 --rounds 5 "r20=GH_CONV_R:20" "r21=GH_CONV_R:21" "r22=GH_CONV_R:22" "r28=GH_CONV_R:28" "r64=GH_CONV_R:64" "r16_gen=GH_CONV_REG:0" 2>&1 | grep -v amdgpu.ids | tail -7
done | tee $O/conv_phases.txt
