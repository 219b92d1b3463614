#!/bin/bash
# round 2, GPU call BH: the band-aliasing rule (GH_BAND_ALIAS_RULE) on and off: plane Sobel and fused kernel over frame shapes
set -o pipefail
O=gpurun_out/r2bh
mkdir -p $O
for wh in "16384 12288" "24576 8192" "12288 16384" "12288 12288" "10240 16384" "20480 8192" "16384 16384" "32768 8192" "16370 16384" "15360 16384" "8192 16384"; do
  set -- $wh
  echo "== sobel $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op sobel --w $1 --h $2 --rounds 5 --in-ring 6 --out-ring 6 "rule_off=GH_BAND_ALIAS_RULE:0" 2>&1 | grep -v amdgpu.ids | tail -2
done | tee $O/sobel_rule.txt
for wh in "12288 16384" "10240 16384" "12288 8192" "8192 8192" "20480 8192" "24576 8192" "16384 16384" "15360 16384"; do
  set -- $wh
  echo "== fused $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op fused --w $1 --h $2 --rounds 5 --in-ring 3 --out-ring 6 "rule_off=GH_BAND_ALIAS_RULE:0" 2>&1 | grep -v amdgpu.ids | tail -2
done | tee $O/fused_rule.txt
