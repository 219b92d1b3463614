#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel statistics and counters of the contract bench command.
#   usage: tools/profile_round.sh <out dir under gpurun_out/>
# The default bench command times the fused pipeline (its `value`) and, in a second timed region, the split pipeline,
# so ONE command covers k_fused_dma, k_gray_rgb_to_plane_w and k_sobel_dma2.  Kernel trace / stats and the PMC passes are
# separate processes (never combined); FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), the SQ / GRBM counters in
# a third.  The program follows `--` directly (no env / bash -c hop).
set -e -o pipefail
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
echo "stats done"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_$ctr" -- python3 bench.py --no-cpu-baseline --no-check > "$OUT/pmc_$ctr.json" 2> "$OUT/pmc_$ctr.err"
  echo "pmc $ctr done"
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/pmc_SQ" -- python3 bench.py --no-cpu-baseline --no-check > "$OUT/pmc_SQ.json" 2> "$OUT/pmc_SQ.err"
echo "pmc SQ done"
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python3 tools/bench_kernels.py > "$OUT/kernel_table.txt" 2>&1
echo "profile round complete"
