#!/bin/bash
# round 2, GPU call L: the driver's round-end sequence on a fresh box: GPU tests, smoke, default bench, two-rank plain command
set -o pipefail
O=gpurun_out/r2l
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
GH_BENCH_REHEARSAL=1 timeout -k 10 300 python3 bench.py --gpus 2 > $O/reh2.json 2> $O/reh2.err; echo "bench --gpus 2 (rehearsal) rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r2l/bench.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], "ceiling", d["copy_ceiling_GBs"], {k: (v["ms"], v["frac_of_8TBs"]) for k, v in d["kernels"].items()})
print(d["parity_check"]); print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
r = json.load(open("gpurun_out/r2l/reh2.json"))
print("reh2", r["ranks"], r["step_pipelined_ms"], r["step_single_frame_ms"], r["step_end_to_end_ms"], r["edge_sum_check"], r["parity_check"])
PY
