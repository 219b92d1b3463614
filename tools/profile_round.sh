#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel statistics and HBM traffic counters of the contract bench.
#   usage: tools/profile_round.sh <out dir under gpurun_out/>
# Kernel trace / stats and the PMC passes are separate processes (never combined), one counter per pass.
set -e -o pipefail
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
for pl in fused split; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$pl" -- python3 bench.py --pipeline $pl --no-cpu-baseline > "$OUT/bench_under_rocprof_$pl.json" 2> "$OUT/stats_$pl.err"
  echo "stats $pl done"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_${ctr}_$pl" -- python3 bench.py --pipeline $pl --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-table --no-check > "$OUT/pmc_${ctr}_$pl.json" 2> "$OUT/pmc_${ctr}_$pl.err"
    echo "pmc $ctr $pl done"
  done
done
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python3 tools/bench_kernels.py > "$OUT/kernel_table.txt" 2>&1
echo "profile round complete"
