#!/bin/bash
# round 2, GPU call U: aligned-down LDS-DMA rows for the fused kernel (and the Sobel again): parity first, then timing
set -o pipefail
O=gpurun_out/r2u
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "widths_that_are_not or shapes or sweep or odd" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for w in 16370 16380 16384; do
  timeout -k 10 200 python3 tools/ab_fused.py --op fused --w $w --h 16384 --rounds 7 --in-ring 3 --out-ring 6 "direct=GH_FUSED_DMA_UNALIGNED:1" "strips=GH_FUSED_DMA_UNALIGNED:0" > $O/fused_$w.txt 2>&1 || exit 1
  echo "== w=$w"; tail -3 $O/fused_$w.txt
done
timeout -k 10 200 python3 tools/ab_fused.py --op fused --w 4099 --h 4099 --rounds 7 --in-ring 3 --out-ring 6 "direct=GH_FUSED_DMA_UNALIGNED:1" "strips=GH_FUSED_DMA_UNALIGNED:0" > $O/fused_4099.txt 2>&1 || exit 1
tail -3 $O/fused_4099.txt
timeout -k 10 200 python3 tools/ab_fused.py --op fused4 --w 16382 --h 8192 --rounds 7 "direct=GH_FUSED_DMA_UNALIGNED:1" "strips=GH_FUSED_DMA_UNALIGNED:0" > $O/fused4_16382.txt 2>&1 || exit 1
tail -3 $O/fused4_16382.txt
