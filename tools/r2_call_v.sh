#!/bin/bash
# round 2, GPU call V: is the "unaligned" penalty really a row-pitch penalty?  aligned widths that are not powers of two
set -o pipefail
O=gpurun_out/r2v
mkdir -p $O
for w in 16384 16368 15360 17408 16370 12288 8192; do
  timeout -k 10 200 python3 tools/ab_fused.py --op fused --w $w --h 16384 --rounds 5 --in-ring 3 --out-ring 6 > $O/fused_$w.txt 2>&1 || exit 1
  timeout -k 10 200 python3 tools/ab_fused.py --op sobel --w $w --h 16384 --rounds 5 --in-ring 6 --out-ring 6 > $O/sobel_$w.txt 2>&1 || exit 1
  echo "== w=$w"; tail -1 $O/fused_$w.txt; tail -1 $O/sobel_$w.txt
done
