#!/bin/bash
# round 2, GPU call D: band-march access-pattern probe vs the strip skeleton (no arithmetic in either)
set -o pipefail
O=gpurun_out/r2d
mkdir -p $O
export GRAYHIP_LIB=$PWD/build_probes/libgrayhip_probes.so
V=("skeleton=GH_SOBEL_DMA:33")
for rw in 2 4 6 8 14; do for sk in 0 1 3 5; do V+=("march${rw}_s${sk}=GH_SOBEL_DMA:$((100+rw)),GH_SOBEL_SKEW:${sk}"); done; done
python3 tools/ab_fused.py --op sobel --rounds 4 "${V[@]}" > $O/ab_march_ring1.txt 2>&1; echo "ring1 rc=$?"
python3 tools/ab_fused.py --op sobel --rounds 4 --in-ring 6 --out-ring 6 "${V[@]}" > $O/ab_march_ring6.txt 2>&1; echo "ring6 rc=$?"
grep -v amdgpu.ids $O/ab_march_ring1.txt $O/ab_march_ring6.txt
