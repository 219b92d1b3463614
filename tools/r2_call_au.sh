#!/bin/bash
# round 2, GPU call AU: k_sobel_dma2 with and without the scheduling barriers between its two strips (two library builds, alternating processes)
set -o pipefail
O=gpurun_out/r2au
mkdir -p $O
for rep in 1 2 3; do
  for lib in prod nobar; do
    if [ $lib = nobar ]; then export GRAYHIP_LIB=$PWD/build_probes/nobar/libgrayhip_nobar.so; else unset GRAYHIP_LIB; fi
    timeout -k 10 200 python3 tools/ab_fused.py --op sobel --rounds 7 --in-ring 6 --out-ring 6 > $O/sobel_${lib}_$rep.txt 2>&1 || exit 1
    echo "$lib rep $rep: $(tail -1 $O/sobel_${lib}_$rep.txt)"
  done
done
