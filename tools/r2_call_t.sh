#!/bin/bash
# round 2, GPU call T: aligned-down LDS-DMA rows for the plane Sobel on frames with unaligned rows -- parity first, then timing
set -o pipefail
O=gpurun_out/r2t
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "sobel_widths_that_are_not or sobel_shapes or sweep" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for w in 16370 16380 16384; do
  timeout -k 10 200 python3 tools/ab_fused.py --op sobel --w $w --h 16384 --rounds 7 --in-ring 6 --out-ring 6 "direct=GH_SOBEL_DMA_UNALIGNED:1" "down_plain=GH_SOBEL_ST:0" "direct_plain=GH_SOBEL_DMA_UNALIGNED:1,GH_SOBEL_ST:0" > $O/sobel_$w.txt 2>&1 || exit 1
  echo "== w=$w"; tail -4 $O/sobel_$w.txt
done
