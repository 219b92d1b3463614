#!/bin/bash
# round 2, GPU call AL: the bench's own in-region Sobel figure with one and two strips per wave, alternating runs on one box
set -o pipefail
O=gpurun_out/r2al
mkdir -p $O
for rep in 1 2 3; do
  for ns in 1 0; do
    GH_SOBEL_NS=$ns timeout -k 10 300 python3 bench.py --cpu-budget-s 1 > $O/bench_ns${ns}_$rep.json 2> $O/bench_ns${ns}_$rep.err || { tail $O/bench_ns${ns}_$rep.err; exit 1; }
    python3 -c "
import json; r = json.load(open('$O/bench_ns${ns}_$rep.json')); k = r['kernels']; print('ns=$ns rep $rep', r['value'], {n: (v.get('ms'), v.get('frac_of_8TBs')) for n, v in k.items() if isinstance(v, dict)})"
  done
done
