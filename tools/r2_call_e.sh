#!/bin/bash
# round 2, GPU call E: RGBA grey kernels (grid forms), full GPU test-suite, profile round with the final kernels
set -o pipefail
O=gpurun_out/r2e
mkdir -p $O
python3 tools/ab_fused.py --op gray4 --rounds 5 "b8192=GH_GRAY_MAX_BLOCKS:8192" "b4096=GH_GRAY_MAX_BLOCKS:4096" "b2048=GH_GRAY_MAX_BLOCKS:2048" "b16384=GH_GRAY_MAX_BLOCKS:16384" > $O/ab_gray4.txt 2>&1
python3 tools/ab_fused.py --op inplace4 --rounds 5 "b8192=GH_GRAY_MAX_BLOCKS:8192" "b4096=GH_GRAY_MAX_BLOCKS:4096" "b2048=GH_GRAY_MAX_BLOCKS:2048" > $O/ab_inplace4.txt 2>&1
python3 tools/ab_fused.py --op gray --rounds 5 "b8192=GH_GRAY_MAX_BLOCKS:8192" "b4096=GH_GRAY_MAX_BLOCKS:4096" "b2048=GH_GRAY_MAX_BLOCKS:2048" > $O/ab_gray3.txt 2>&1
grep -v amdgpu.ids $O/ab_gray4.txt $O/ab_inplace4.txt $O/ab_gray3.txt
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
bash tools/profile_round.sh gpurun_out/r2e/prof; echo "profile rc=$?"
