#!/bin/bash
# round 2, GPU call BM: regular-band kernels with the last band shifted up (frames whose rows do not split into whole bands): suite, stress, shapes
set -o pipefail
O=gpurun_out/r2bm
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
timeout -k 10 300 python3 tools/stress_parity.py 120 31337 > $O/stress.log 2>&1 || { tail $O/stress.log; exit 1; }
tail -1 $O/stress.log
for wh in "1920 1080" "5000 3000" "7680 4320" "2037 1362" "12288 8200" "16384 16390"; do
  set -- $wh
  echo "== sobel $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op sobel --w $1 --h $2 --rounds 7 --in-ring 6 --out-ring 6 "generic=GH_SOBEL_REG:0,GH_SOBEL_NS:1" 2>&1 | grep -v amdgpu.ids | tail -2
  echo "== fused $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op fused --w $1 --h $2 --rounds 7 --in-ring 3 --out-ring 6 "generic=GH_FUSED_REG:0" 2>&1 | grep -v amdgpu.ids | tail -2
done | tee $O/shapes.txt
