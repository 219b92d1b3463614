#!/bin/bash
# round 2, GPU call J: aligned-down DMA variant of the plane Sobel (parity first, then timing on odd widths)
set -o pipefail
O=gpurun_out/r2j
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -x -q -k "sobel_shapes or not_multiples or any_width or geometry_sweep or unaligned_device or latency_noise or row_bands" > $O/pytest_unaligned.log 2>&1 || { echo "unaligned tests failed"; tail -30 $O/pytest_unaligned.log; exit 1; }
tail -2 $O/pytest_unaligned.log
for geo in "16370 16384" "4099 4099" "2037 1362"; do
  set -- $geo
  echo "== sobel $1 x $2"
  python3 tools/ab_fused.py --op sobel --w $1 --h $2 --rounds 5 "direct=GH_SOBEL_DMA_BLOCKS:0" "strips_tail=GH_SOBEL_DMA_UNALIGNED:0,GH_SOBEL_UNALIGNED_MIN_PX:1" 2>&1 | grep -v amdgpu.ids
done | tee $O/ab_sobel_blocks.txt
python3 tools/ab_fused.py --op sobel --rounds 5 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_sobel_blocks.txt
