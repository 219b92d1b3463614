#!/usr/bin/env python3
"""usage: pmc_medians.py <dir with rocprofv3 counter_collection.csv files> -- median of every counter per kernel (largest grid)."""
import collections, csv, glob, re, statistics, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        if m:
            acc[m.group(1)][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
for k in sorted(acc):
    for c in sorted(acc[k]):
        s = acc[k][c]
        g = max(x for x, _ in s)
        v = [y for x, y in s if x == g]
        print(f"{k:18s} {c:42s} n={len(v):3d} median {statistics.median(v):16.1f}")
