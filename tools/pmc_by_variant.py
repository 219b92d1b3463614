#!/usr/bin/env python3
"""usage: pmc_by_variant.py <dir with rocprofv3 counter_collection.csv files> -- median of every counter per kernel INSTANTIATION (template
arguments kept: the variants of one kernel are different symbols), largest grid only, one line per kernel, counters as columns."""
import collections, csv, glob, re, statistics, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if m:
            acc[m.group(1).replace(" ", "")][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
out = {}
for k in sorted(acc):
    out[k] = {}
    for c in sorted(acc[k]):
        s = acc[k][c]
        g = max(x for x, _ in s)
        out[k][c] = statistics.median([y for x, y in s if x == g])
if "--json" in sys.argv:
    import json
    print(json.dumps(out, indent=1))
else:
    for k, d in out.items():
        print(k)
        for c, v in d.items():
            print(f"    {c:40s} {v:18.1f}")
