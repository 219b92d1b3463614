#!/bin/bash
# round 2, GPU call AJ: two-strip regular-band Sobel kernel across frame shapes (band height 64 / 16) against the one-strip kernel
set -o pipefail
O=gpurun_out/r2aj
mkdir -p $O
for wh in "16384 4096" "8192 8192" "32768 8192" "4096 16384" "2048 2048" "4096 4096" "16384 1024" "7680 4352"; do
set -- $wh
timeout -k 10 300 python3 tools/ab_fused.py --op sobel --w $1 --h $2 --rounds 7 --in-ring 6 --out-ring 6 "ns2_r64=GH_SOBEL_NS:2,GH_SOBEL_NS_D:33,GH_SOBEL_R:64" "ns2_r16=GH_SOBEL_NS:2,GH_SOBEL_NS_D:33,GH_SOBEL_R:16" > $O/sobel_$1x$2.txt 2>&1 || { tail $O/sobel_$1x$2.txt; exit 1; }
echo "== $1 x $2"; tail -3 $O/sobel_$1x$2.txt
done
