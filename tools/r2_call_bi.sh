#!/bin/bash
# round 2, GPU call BI: fused kernel, band heights 16 / 32 / 64 on the shapes where the aliasing rule fires
set -o pipefail
O=gpurun_out/r2bi
mkdir -p $O
for wh in "24576 8192" "20480 8192" "12288 16384" "10240 16384" "8192 16384" "12288 12288" "20480 12288"; do
  set -- $wh
  echo "== fused $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op fused --w $1 --h $2 --rounds 7 --in-ring 3 --out-ring 6 "r16=GH_FUSED_DMA_R:16" "r32=GH_FUSED_DMA_R:32" "r64=GH_FUSED_DMA_R:64" 2>&1 | grep -v amdgpu.ids | tail -4
done | tee $O/fused_r_alias.txt
