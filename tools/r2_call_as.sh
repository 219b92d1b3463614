#!/bin/bash
# round 2, GPU call AS: bench with region event pairs (no per-step events in one-kernel regions): default, split-first, two-rank rehearsals
set -o pipefail
O=gpurun_out/r2as
mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
timeout -k 10 400 python3 bench.py --pipeline split --cpu-budget-s 1 > $O/bench_split.json 2> $O/bench_split.err || { tail $O/bench_split.err; exit 1; }
GH_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 50 > $O/reh2.json 2> $O/reh2.err || { tail $O/reh2.err; exit 1; }
GH_BENCH_REHEARSAL=loopback timeout -k 10 300 python3 bench.py --gpus 2 --steps 50 > $O/reh2_lb.json 2> $O/reh2_lb.err || { tail $O/reh2_lb.err; exit 1; }
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "bench or plain_command" > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
python3 - <<'PY'
import json
for f in ("bench", "bench_split"):
    r = json.load(open(f"gpurun_out/r2as/{f}.json"))
    print(f, r["value"], r["ms_per_step"], r["ms_per_step_events"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"], r["parity_check"][:30])
    print("   ", {k: (v["ms"], v["frac_of_8TBs"], v["timed_region"]) for k, v in r["kernels"].items()})
    print("   ", r["second_timed_region"], r["third_timed_region"])
for f in ("reh2", "reh2_lb"):
    r = json.load(open(f"gpurun_out/r2as/{f}.json"))
    print(f, r["ranks"], r["rccl_ranks"], "pipelined", r["step_pipelined_ms"], "single", r["step_single_frame_ms"], "e2e", r["step_end_to_end_ms"], r["ms_per_step_events"], r["edge_sum_check"][:20])
PY
