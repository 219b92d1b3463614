#!/bin/bash
# round 2, GPU call BF: fused band-height rule with the balance term: the shapes it changes and the ones it must not change
set -o pipefail
O=gpurun_out/r2bf
mkdir -p $O
for wh in "12288 8192" "10240 10240" "16384 4096" "16384 8192" "16384 3072" "16384 2048" "8192 4096" "16384 6144" "14336 6144" "16384 16384"; do
  set -- $wh
  echo "== fused $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op fused --w $1 --h $2 --rounds 5 --in-ring 4 --out-ring 6 "r16=GH_FUSED_DMA_R:16" "r32=GH_FUSED_DMA_R:32" "r64=GH_FUSED_DMA_R:64" 2>&1 | grep -v amdgpu.ids | tail -4
done | tee $O/fused_rule.txt
