#!/bin/bash
# round 2, GPU call G: fewer concurrent row streams at equal bytes in flight (residency x ring depth)
set -o pipefail
O=gpurun_out/r2g
mkdir -p $O
V=()
for pad in 0 36 60 100; do for d in 3 4 6 8; do V+=("pad${pad}_d${d}=GH_SOBEL_PAD_KB:${pad},GH_SOBEL_DMA:${d}"); done; done
python3 tools/ab_fused.py --op sobel --rounds 4 --in-ring 4 --out-ring 4 "${V[@]}" > $O/ab_sobel_streams.txt 2>&1; echo "rc=$?"
V=()
for pad in 0 30 60 110; do for d in 2 3 4; do V+=("pad${pad}_d${d}=GH_FUSED_DMA_PAD_KB:${pad},GH_FUSED_DMA:${d}"); done; done
python3 tools/ab_fused.py --op fused --rounds 4 --in-ring 3 --out-ring 4 "${V[@]}" > $O/ab_fused_streams.txt 2>&1; echo "rc=$?"
grep -v amdgpu.ids $O/ab_sobel_streams.txt $O/ab_fused_streams.txt
