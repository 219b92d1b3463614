#!/bin/bash
# round 2, GPU call BN: shifted last band only where it does not overflow the resident slots: ragged-rows tests, shapes on both sides of the rule
set -o pipefail
O=gpurun_out/r2bn
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for wh in "16384 16390" "16384 8200" "12288 8200" "4096 4100" "16384 4100" "7680 4320"; do
  set -- $wh
  echo "== sobel $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op sobel --w $1 --h $2 --rounds 5 --in-ring 6 --out-ring 6 "generic=GH_SOBEL_REG:0,GH_SOBEL_NS:1" 2>&1 | grep -v amdgpu.ids | tail -2
done | tee $O/shapes.txt
