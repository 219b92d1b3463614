#!/bin/bash
# round 2, GPU call AK: two-strip Sobel kernel as the default for large aligned frames -- parity tests, stress, full suite, bench
set -o pipefail
O=gpurun_out/r2ak
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "two_strips" > $O/pytest_ns2.log 2>&1 || { tail -30 $O/pytest_ns2.log; exit 1; }
tail -2 $O/pytest_ns2.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
python3 -c "
import json; r = json.load(open('$O/bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['parity_check'][:40], {k: v.get('frac_of_8TBs') for k, v in r['kernels'].items() if isinstance(v, dict)}); print(r.get('second_timed_region'))"
