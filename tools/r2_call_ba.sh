#!/bin/bash
# round 2, GPU call BA: regular-band form of the fused kernel on 16-row bands: mid-size / small frames and the shard sizes
set -o pipefail
O=gpurun_out/r2ba
mkdir -p $O
for wh in "16384 2048" "16384 1024" "7680 4352" "4096 4096" "3840 2160" "1920 1088" "12288 4096" "5000 3008" "8192 2048" "16384 3072"; do
  set -- $wh
  echo "== fused $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op fused --w $1 --h $2 --rounds 7 --in-ring 4 --out-ring 6 "generic=GH_FUSED_REG:0" 2>&1 | grep -v amdgpu.ids | tail -2
done | tee $O/fused_reg16.txt
