#!/bin/bash
# round 2, GPU call AC: loader-wave skeleton (wave 0 issues the workgroup's 4 KiB per row, one barrier per row, four waves store 1 KiB each)
set -o pipefail
O=gpurun_out/r2ac
mkdir -p $O
export GRAYHIP_LIB=$PWD/build_probes/libgrayhip_probes.so
timeout -k 10 300 python3 tools/ab_fused.py --op sobel --rounds 7 --in-ring 6 --out-ring 6 "skeleton=GH_SOBEL_DMA:33" "skeleton_barrier=GH_SOBEL_DMA:53" \
  "w4_d3=GH_SOBEL_DMA:403" "loader_d3=GH_SOBEL_DMA:1703" "loader_d4=GH_SOBEL_DMA:1704" "loader_d4_o0=GH_SOBEL_DMA:1704,GH_SOBEL_ORDER:0" "loader_d4_r32=GH_SOBEL_DMA:1704,GH_SOBEL_R:32" > $O/sobel.txt 2>&1 || { tail $O/sobel.txt; exit 1; }
tail -8 $O/sobel.txt
