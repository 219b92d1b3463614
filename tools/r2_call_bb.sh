#!/bin/bash
# round 2, GPU call BB: regular-band form of the convolution's ring kernel: parity, then shapes
set -o pipefail
O=gpurun_out/r2bb
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_old_api.py -x -q -k "conv or old" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for wh in "16384 16384" "16368 16384" "7680 4352" "4096 4096" "3840 2160" "1920 1088" "16384 2048" "12288 4096"; do
  set -- $wh
  echo "== conv $1 x $2"
  timeout -k 10 300 python3 tools/ab_fused.py --op conv --w $1 --h $2 --rounds 5 "generic=GH_CONV_REG:0" "reg_r16=GH_CONV_R:16" "generic_r16=GH_CONV_REG:0,GH_CONV_R:16" 2>&1 | grep -v amdgpu.ids | tail -4
done | tee $O/conv_reg.txt
