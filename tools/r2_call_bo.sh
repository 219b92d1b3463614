#!/bin/bash
# round 2, GPU call BO: fused kernel on mid-size frames: ring depth 2 with / without the residency pad, depth 3 (the current choice is best or within 2 % everywhere but 7680 x 4352; unchanged)
set -o pipefail
O=gpurun_out/r2bo
mkdir -p $O
for wh in "7680 4352" "12288 8192" "10240 10240" "16384 6144" "14336 6144" "20480 8192" "16384 12288" "12288 12288"; do
  set -- $wh
  echo "== fused $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op fused --w $1 --h $2 --rounds 5 --in-ring 4 --out-ring 6 "d2_pad0=GH_FUSED_DMA:2,GH_FUSED_DMA_PAD_KB:0" "d2_pad30=GH_FUSED_DMA:2,GH_FUSED_DMA_PAD_KB:30" "d3_pad0=GH_FUSED_DMA:3,GH_FUSED_DMA_PAD_KB:0" "d3_pad30=GH_FUSED_DMA:3,GH_FUSED_DMA_PAD_KB:30" 2>&1 | grep -v amdgpu.ids | tail -5
done | tee $O/fused_pad.txt
