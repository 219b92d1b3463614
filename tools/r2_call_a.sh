#!/bin/bash
# round 2, GPU call A: GPU test-suite, the new bench (N=1, N=2 rehearsals), Sobel / fused band-height x order exploration
set -o pipefail
O=gpurun_out/r2a
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
GH_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 20 --warmup 2 > $O/reh2.json 2> $O/reh2.err; echo "reh2 rc=$?"
GH_BENCH_REHEARSAL=loopback timeout -k 10 300 python3 bench.py --gpus 2 --steps 20 --warmup 2 > $O/reh2_lb.json 2> $O/reh2_lb.err; echo "reh2_lb rc=$?"
export GRAYHIP_LIB=$PWD/build_probes/libgrayhip_probes.so
# plane Sobel: small bands + XCD-major order (each XCD streams its own slab of the frame), default and nt DMA policy
python3 tools/ab_fused.py --op sobel --rounds 5 \
  "R8_xcd=GH_SOBEL_R:8,GH_SOBEL_ORDER:4" "R16_xcd=GH_SOBEL_R:16,GH_SOBEL_ORDER:4" "R32_xcd=GH_SOBEL_R:32,GH_SOBEL_ORDER:4" "R64_xcd=GH_SOBEL_R:64,GH_SOBEL_ORDER:4" \
  "R8_plain=GH_SOBEL_R:8,GH_SOBEL_ORDER:0" "R16_plain=GH_SOBEL_R:16,GH_SOBEL_ORDER:0" "R4_xcd=GH_SOBEL_R:4,GH_SOBEL_ORDER:4" \
  "R8_xcd_lp0=GH_SOBEL_R:8,GH_SOBEL_ORDER:4,GH_SOBEL_DMA:23" "R16_xcd_lp0=GH_SOBEL_R:16,GH_SOBEL_ORDER:4,GH_SOBEL_DMA:23" "R64_lp0=GH_SOBEL_DMA:23" \
  "R8_xcd_2wg=GH_SOBEL_R:8,GH_SOBEL_ORDER:4,GH_SOBEL_PAD_KB:60" "R16_xcd_2wg=GH_SOBEL_R:16,GH_SOBEL_ORDER:4,GH_SOBEL_PAD_KB:60" "R64_2wg=GH_SOBEL_PAD_KB:60" \
  "R8_xcd_3wg=GH_SOBEL_R:8,GH_SOBEL_ORDER:4,GH_SOBEL_PAD_KB:36" "R64_3wg=GH_SOBEL_PAD_KB:36" "nosqrt=GH_SOBEL_DMA:13" "skeleton=GH_SOBEL_DMA:33" "R8_xcd_skeleton=GH_SOBEL_R:8,GH_SOBEL_ORDER:4,GH_SOBEL_DMA:33" \
  > $O/ab_sobel_smallR.txt 2>&1; echo "ab sobel rc=$?"
python3 tools/ab_fused.py --op fused --rounds 5 \
  "R8_xcd=GH_FUSED_DMA_R:8,GH_FUSED_DMA_ORDER:4" "R16_xcd=GH_FUSED_DMA_R:16,GH_FUSED_DMA_ORDER:4" "R32_xcd=GH_FUSED_DMA_R:32,GH_FUSED_DMA_ORDER:4" \
  "R8_plain=GH_FUSED_DMA_R:8,GH_FUSED_DMA_ORDER:0" "R4_xcd=GH_FUSED_DMA_R:4,GH_FUSED_DMA_ORDER:4" "R16_6=GH_FUSED_DMA_R:16,GH_FUSED_DMA_ORDER:6" \
  "R8_xcd_lp0=GH_FUSED_DMA_R:8,GH_FUSED_DMA_ORDER:4,GH_FUSED_DMA:2,GH_FUSED_DMA_ST:102,GH_FUSED_DMA_PAD_KB:30" \
  "R64_lp0=GH_FUSED_DMA:2,GH_FUSED_DMA_ST:102,GH_FUSED_DMA_PAD_KB:30" "skeleton=GH_FUSED_DMA:3,GH_FUSED_DMA_ST:12" \
  > $O/ab_fused_smallR.txt 2>&1; echo "ab fused rc=$?"
unset GRAYHIP_LIB
python3 tools/bench_kernels.py > $O/kernel_table.txt 2>&1; echo "kernels rc=$?"
cat $O/ab_sobel_smallR.txt $O/ab_fused_smallR.txt $O/kernel_table.txt
