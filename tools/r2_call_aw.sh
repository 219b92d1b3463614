#!/bin/bash
# round 2, GPU call AW: bench N > 1 with the batched halo prefetch (ring of 8): two- and four-rank rehearsals (gloo exchange and native loopback), with and without batching
set -o pipefail
O=gpurun_out/r2aw
mkdir -p $O
GH_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 50 > $O/reh2.json 2> $O/reh2.err || { tail $O/reh2.err; exit 1; }
GH_BENCH_REHEARSAL=loopback timeout -k 10 300 python3 bench.py --gpus 2 --steps 100 > $O/reh2_lb.json 2> $O/reh2_lb.err || { tail $O/reh2_lb.err; exit 1; }
GH_BENCH_REHEARSAL=loopback timeout -k 10 300 python3 bench.py --gpus 2 --steps 100 --halo-batch 0 > $O/reh2_lb_nobatch.json 2> $O/reh2_lb_nobatch.err || { tail $O/reh2_lb_nobatch.err; exit 1; }
GH_BENCH_REHEARSAL=loopback timeout -k 10 300 python3 bench.py --gpus 4 --steps 100 > $O/reh4_lb.json 2> $O/reh4_lb.err || { tail $O/reh4_lb.err; exit 1; }
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "plain_command or loopback or sharded" > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
python3 - <<'PY'
import json
for f in ("reh2", "reh2_lb", "reh2_lb_nobatch", "reh4_lb"):
    r = json.load(open(f"gpurun_out/r2aw/{f}.json"))
    print(f, r["ranks"], r["rccl_ranks"], "pipelined", r["step_pipelined_ms"], "single", r["step_single_frame_ms"], "e2e", r["step_end_to_end_ms"], "host", r["host_enqueue_us_per_step"], r["edge_sum_check"][:20], r["parity_check"][:14])
    print("   ", r["halo_exchange"][:200])
PY
