#!/bin/bash
# round 2, GPU call AT: outer-product (separable) form of the 3x3 convolution (6 instead of 11 packed operations per byte pair): parity green, 0.2949 vs 0.2964 ms -- the kernel is not VALU-bound; not kept
set -o pipefail
O=gpurun_out/r2at
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_old_api.py -x -q -k "conv or old" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for w in 16384 16370; do
  timeout -k 10 200 python3 tools/ab_fused.py --op conv --w $w --h 16384 --rounds 7 "generic=GH_CONV_SEP:0" > $O/conv_$w.txt 2>&1 || { tail $O/conv_$w.txt; exit 1; }
  echo "== w=$w"; tail -2 $O/conv_$w.txt
done
