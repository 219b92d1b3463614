#!/bin/bash
# round 2, GPU call Y: row-synchronous workgroups (one s_barrier per row step: the four strips of a workgroup read 4 KiB of a row together)
set -o pipefail
O=gpurun_out/r2y
mkdir -p $O
export GRAYHIP_LIB=$PWD/build_probes/libgrayhip_probes.so
timeout -k 10 300 python3 tools/ab_fused.py --op sobel --rounds 7 --in-ring 6 --out-ring 6 "barrier=GH_SOBEL_DMA:43" "skeleton=GH_SOBEL_DMA:33" "skeleton_barrier=GH_SOBEL_DMA:53" > $O/sobel.txt 2>&1 || { tail $O/sobel.txt; exit 1; }
tail -5 $O/sobel.txt
