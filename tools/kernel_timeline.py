#!/usr/bin/env python3
"""Prints the last N kernel dispatches of a rocprofv3 --kernel-trace CSV as a timeline (us, relative to the first one shown).  Developer tool."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} .. {e:9.1f} us ({e - s:6.1f})  queue {r.get('Queue_Id', '?'):>3s}  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>8s}  {r['Kernel_Name'][:90]}")
