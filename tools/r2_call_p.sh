#!/bin/bash
# round 2, GPU call P: serial sharded step as default -- loopback parity tests, both bench rehearsal modes
set -o pipefail
O=gpurun_out/r2p
mkdir -p $O
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "native or sharded or plain_command or loopback" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
GH_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 50 > $O/reh2.json 2> $O/reh2.err; echo "reh2 rc=$?"
GH_BENCH_REHEARSAL=loopback timeout -k 10 300 python3 bench.py --gpus 2 --steps 50 > $O/reh2_lb.json 2> $O/reh2_lb.err; echo "reh2_lb rc=$?"
GH_BENCH_REHEARSAL=loopback timeout -k 10 300 python3 bench.py --gpus 4 --steps 50 > $O/reh4_lb.json 2> $O/reh4_lb.err; echo "reh4_lb rc=$?"
python3 - <<'PY'
import json
for f in ("reh2", "reh2_lb", "reh4_lb"):
    r = json.load(open(f"gpurun_out/r2p/{f}.json"))
    print(f, r["ranks"], r["rccl_ranks"], "pipelined", r["step_pipelined_ms"], "single", r["step_single_frame_ms"], "e2e", r["step_end_to_end_ms"], r["edge_sum_check"][:20], r["parity_check"][:12])
PY
