#!/bin/bash
# round 2, GPU call AZ: the regular one-strip form as default for 16-row bands, two strips only on widths that are multiples of 8192: full GPU suite, then shapes
set -o pipefail
O=gpurun_out/r2az
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for wh in "12288 8192" "7680 4352" "4096 4096" "16384 1024" "3840 2160" "1920 1088" "16384 4096" "16384 16384" "16370 16384" "5000 3008" "2037 1360"; do
  set -- $wh
  echo "== sobel $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op sobel --w $1 --h $2 --rounds 7 --in-ring 6 --out-ring 6 "no_reg=GH_SOBEL_REG:0" "round2_mid=GH_SOBEL_REG:0,GH_SOBEL_NS:1" 2>&1 | grep -v amdgpu.ids | tail -3
done | tee $O/sobel_shapes.txt
