#!/bin/bash
# round 2, GPU call BG: plane Sobel band heights where the workgroup count is not a multiple of the CU count
set -o pipefail
O=gpurun_out/r2bg
mkdir -p $O
for wh in "16384 12288" "24576 8192" "12288 16384" "12288 12288" "10240 16384" "16384 6144" "20480 8192"; do
  set -- $wh
  echo "== sobel $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op sobel --w $1 --h $2 --rounds 5 --in-ring 6 --out-ring 6 "r16=GH_SOBEL_R:16" "r64=GH_SOBEL_R:64" "one_r16=GH_SOBEL_NS:1,GH_SOBEL_R:16" "one_r64=GH_SOBEL_NS:1,GH_SOBEL_R:64" 2>&1 | grep -v amdgpu.ids | tail -5
done | tee $O/sobel_r.txt
