#!/bin/bash
# round 2, GPU call Q: where do frames with unaligned rows lose their 8 % -- the DMA loads or the (partial-line) stores?
# plane Sobel on 16370- / 16376- / 16384-wide frames, store policy nt / plain / sc1, inputs and outputs rotating over 6 buffers
set -o pipefail
O=gpurun_out/r2q
mkdir -p $O
for w in 16384 16370 16376 16380; do
  timeout -k 10 200 python3 tools/ab_fused.py --op sobel --w $w --h 16384 --rounds 7 --in-ring 6 --out-ring 6 "plain=GH_SOBEL_ST:0" "sc1=GH_SOBEL_ST:1" "strips=GH_SOBEL_DMA_UNALIGNED:0" > $O/sobel_$w.txt 2>&1 || exit 1
  echo "== w=$w"; tail -5 $O/sobel_$w.txt
done
