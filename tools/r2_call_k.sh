#!/bin/bash
# round 2, GPU call K: the same 268 Mpixel of plane Sobel / fused work at different row pitches (1 KiB ... 64 KiB rows)
set -o pipefail
O=gpurun_out/r2k
mkdir -p $O
for geo in "1024 262144" "2048 131072" "4096 65536" "8192 32768" "16384 16384" "32768 8192" "65536 4096"; do
  set -- $geo
  echo "== sobel $1 x $2"; python3 tools/ab_fused.py --op sobel --w $1 --h $2 --rounds 4 --in-ring 3 --out-ring 3 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== fused $1 x $2"; python3 tools/ab_fused.py --op fused --w $1 --h $2 --rounds 4 --in-ring 2 --out-ring 3 2>&1 | grep -v amdgpu.ids | tail -1
done | tee $O/pitch_sweep.txt
