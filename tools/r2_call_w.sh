#!/bin/bash
# round 2, GPU call W: 16370-wide frames (unaligned rows) against 16368-wide ones (16-byte aligned rows, same non-power-of-two pitch
# class) and 16384, alternating processes on one box
set -o pipefail
O=gpurun_out/r2w
mkdir -p $O
for rep in 1 2 3; do
  for w in 16368 16370 16384; do
    timeout -k 10 200 python3 tools/ab_fused.py --op fused --w $w --h 16384 --rounds 5 --in-ring 3 --out-ring 6 "direct=GH_FUSED_DMA_UNALIGNED:1" > $O/fused_${w}_$rep.txt 2>&1 || exit 1
    timeout -k 10 200 python3 tools/ab_fused.py --op sobel --w $w --h 16384 --rounds 5 --in-ring 6 --out-ring 6 "direct=GH_SOBEL_DMA_UNALIGNED:1" "plain=GH_SOBEL_ST:0" > $O/sobel_${w}_$rep.txt 2>&1 || exit 1
    echo "== rep $rep w=$w"; tail -2 $O/fused_${w}_$rep.txt; tail -3 $O/sobel_${w}_$rep.txt
  done
done
