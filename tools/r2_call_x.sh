#!/bin/bash
# round 2, GPU call X: full GPU suite + smoke + bench after the aligned-down DMA rows became the default for unaligned frames
set -o pipefail
O=gpurun_out/r2x
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
python3 -c "
import json; r = json.load(open('$O/bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['parity_check'][:40], {k: v.get('frac_of_8TBs') for k, v in r['kernels'].items() if isinstance(v, dict)})"
