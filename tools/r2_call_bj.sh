#!/bin/bash
# round 2, GPU call BJ: after the aliasing rule and the balance terms: full GPU suite, bench, a shape sweep of both kernels against the old rules
set -o pipefail
O=gpurun_out/r2bj
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
timeout -k 10 400 python3 bench.py --cpu-budget-s 1 > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
python3 -c "
import json; r = json.load(open('$O/bench.json')); print(r['value'], r['roofline']['frac'], {k: v['frac_of_8TBs'] for k, v in r['kernels'].items()}, r['parity_check'][:12])"
for wh in "16384 12288" "16384 10240" "24576 8192" "12288 12288" "16384 16384"; do
  set -- $wh
  echo "== sobel $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op sobel --w $1 --h $2 --rounds 5 --in-ring 6 --out-ring 6 "r64=GH_SOBEL_R:64" "r16=GH_SOBEL_R:16" 2>&1 | grep -v amdgpu.ids | tail -3
done | tee $O/sobel_final.txt
