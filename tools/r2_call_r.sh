#!/bin/bash
# round 2, GPU call R: fused kernel on frames with unaligned rows -- store policy nt / plain / sc1 (probes build), rotating buffers
set -o pipefail
O=gpurun_out/r2r
mkdir -p $O
export GRAYHIP_LIB=$PWD/build_probes/libgrayhip_probes.so
for w in 16384 16370 16380; do
  timeout -k 10 200 python3 tools/ab_fused.py --op fused --w $w --h 16384 --rounds 7 --in-ring 3 --out-ring 6 "plain=GH_FUSED_DMA_ST:0" "sc1=GH_FUSED_DMA_ST:1" "strips=GH_FUSED_DMA_UNALIGNED:0" > $O/fused_$w.txt 2>&1 || exit 1
  echo "== w=$w"; tail -5 $O/fused_$w.txt
done
