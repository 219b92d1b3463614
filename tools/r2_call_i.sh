#!/bin/bash
# round 2, GPU call I: HBM read traffic of the ring kernels on a frame whose rows are not 16-byte aligned (16370 x 16384)
set -o pipefail
O=gpurun_out/r2i
mkdir -p $O
export TMPDIR=/tmp
for op in fused sobel; do
  for w in 16370 16384; do
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${op}_$w -- python3 tools/ab_fused.py --op $op --w $w --h 16384 --rounds 1 --reps 10 > $O/run_${op}_$w.txt 2>&1
    echo "$op $w done"
  done
done
python3 - <<'PY'
import csv, glob, re, statistics
for d in sorted(glob.glob("gpurun_out/r2i/pmc_*")):
    vals = {}
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_\w+_dma)", r["Kernel_Name"])
            if m and r["Counter_Name"] == "FETCH_SIZE":
                vals.setdefault((m.group(1), int(r["Grid_Size"])), []).append(float(r["Counter_Value"]))
    for (k, g), v in vals.items():
        print(d.split("/")[-1], k, "grid", g, "launches", len(v), "median FETCH_SIZE x2 = %.1f MB" % (statistics.median(v) * 2 * 1024 / 1e6))
PY
