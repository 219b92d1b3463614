#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (developer tool)."""
import collections, csv, glob, re, sys
for d in sys.argv[1:]:
    f = glob.glob(f'{d}/*/*_counter_collection.csv')[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); meta = {}
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_\w+(<[^>]*>)?)', r['Kernel_Name'])
        if not m: continue
        k = m.group(1)
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        meta[k] = f"vgpr={r['VGPR_Count']} lds={r['LDS_Block_Size']} grid={r['Grid_Size']}"
    for k, c in agg.items():
        a = {n: sum(v) / len(v) for n, v in c.items()}
        wc = a.get('SQ_WAVE_CYCLES', 0)
        parts = [f"{n}={v:.4g}" + (f"({v/wc:.2f}/wc)" if wc and n.startswith('SQ_') and n != 'SQ_WAVE_CYCLES' else '') for n, v in sorted(a.items())]
        print(k, meta[k], ' '.join(parts))
