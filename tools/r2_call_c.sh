#!/bin/bash
# round 2, GPU call C: GPU test-suite (unaligned DMA path, stress), odd-width timings, profile round
set -o pipefail
O=gpurun_out/r2c
mkdir -p $O
# the paths that changed since the last green run first, on their own: a fault here must not be followed by further GPU work
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sobel_shapes or not_multiples or any_width or geometry_sweep" > $O/pytest_unaligned.log 2>&1 || { echo "unaligned tests failed"; tail -30 $O/pytest_unaligned.log; exit 1; }
tail -2 $O/pytest_unaligned.log
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
[ "$(tail -1 $O/pytest_gpu.log | grep -c passed)" = 1 ] || exit 1
for geo in "16370 16384" "16384 16384" "2037 1362" "1920 1080" "2000 1333" "4099 4099"; do
  set -- $geo
  echo "== sobel $1 x $2" >> $O/ab_unaligned.txt
  python3 tools/ab_fused.py --op sobel --w $1 --h $2 --rounds 5 "dma_unaligned_off=GH_SOBEL_DMA_UNALIGNED:0,GH_SOBEL_UNALIGNED_MIN_PX:1" "per_pixel=GH_SOBEL_DMA_UNALIGNED:0,GH_SOBEL_UNALIGNED:0" >> $O/ab_unaligned.txt 2>&1
  echo "== fused $1 x $2" >> $O/ab_unaligned.txt
  python3 tools/ab_fused.py --op fused --w $1 --h $2 --rounds 5 "dma_unaligned_off=GH_FUSED_DMA_UNALIGNED:0,GH_FUSED_UNALIGNED_MIN_PX:1" "per_pixel=GH_FUSED_DMA_UNALIGNED:0,GH_FUSED_UNALIGNED_MIN_PX:1073741824" >> $O/ab_unaligned.txt 2>&1
  echo "geo $geo done"
done
grep -v amdgpu.ids $O/ab_unaligned.txt
bash tools/profile_round.sh gpurun_out/r2c/prof; echo "profile rc=$?"
