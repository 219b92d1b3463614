#!/bin/bash
# round 2, GPU call AN: two strips per wave on frames with unaligned rows: parity, then timing (16370 / 16368 / 16380 wide)
set -o pipefail
O=gpurun_out/r2an
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "two_strips or widths_that_are_not or every_base" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for w in 16370 16368 16380; do
  timeout -k 10 200 python3 tools/ab_fused.py --op sobel --w $w --h 16384 --rounds 7 --in-ring 6 --out-ring 6 "one_strip=GH_SOBEL_NS:1" > $O/sobel_$w.txt 2>&1 || exit 1
  echo "== w=$w"; tail -2 $O/sobel_$w.txt
done
