#!/bin/bash
# round 2, GPU call AE: k_conv3x3_dma -- parity (any width / alignment), then timing against the register-staged strips
set -o pipefail
O=gpurun_out/r2ae
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_old_api.py -x -q -k "conv or old" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for w in 16384 16368 16370; do
  timeout -k 10 200 python3 tools/ab_fused.py --op conv --w $w --h 16384 --rounds 5 "strips=GH_CONV_DMA:0" "r32=GH_CONV_R:32" "o0=GH_CONV_ORDER:0" "o8=GH_CONV_ORDER:8" > $O/conv_$w.txt 2>&1 || { tail $O/conv_$w.txt; exit 1; }
  echo "== w=$w"; tail -5 $O/conv_$w.txt
done
