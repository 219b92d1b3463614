#!/usr/bin/env python3
"""Turns the raw output of tools/profile_round.sh (merged back under gpurun_out/) into the committed summaries under
profiles/<name>/ and points profiles/LATEST at them.   usage: make_profile_summary.py gpurun_out/prof profiles/r1_final"""
import collections, csv, glob, json, os, re, shutil, sys
src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
ALGO = {"k_fused_dma": 4, "k_fused_stream": 4, "k_gray_rgb_to_plane_w": 4, "k_sobel_dma": 2, "k_sobel_stream": 2}
NPIX = 16384 * 16384

def kname(s):
    m = re.search(r"(k_\w+)", s)
    return m.group(1) if m else None

for pl in ("fused", "split"):
    f = glob.glob(f"{src}/stats_{pl}/*/*kernel_stats.csv")
    if f:
        shutil.copy(f[0], f"{dst}/kernel_stats_{pl}.csv")
    j = f"{src}/bench_under_rocprof_{pl}.json"
    if os.path.exists(j):
        shutil.copy(j, f"{dst}/bench_under_rocprof_{pl}.json")
for extra in ("bench_default.json", "kernel_table.txt"):
    if os.path.exists(f"{src}/{extra}"):
        shutil.copy(f"{src}/{extra}", f"{dst}/{extra}")

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for pl in ("fused", "split"):
        for f in glob.glob(f"{src}/pmc_{ctr}_{pl}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = kname(r["Kernel_Name"])
                if k in ALGO and r["Counter_Name"] == ctr:
                    acc[k][ctr].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --pipeline {fused,split}` on one "
               "MI355X (tools/profile_round.sh); 16384x16384 RGB8; per-launch averages over the launches that cover the whole frame; "
               "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md section HBM)", "kernels": {}}
for k, c in acc.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    # bench.py also launches these kernels on small check slices; keep the launches near the maximum (whole frame)
    fmax, wmax = max(c["FETCH_SIZE"]), max(c["WRITE_SIZE"])
    fv = [v for v in c["FETCH_SIZE"] if v > 0.8 * fmax]
    wv = [v for v in c["WRITE_SIZE"] if v > 0.8 * wmax]
    fk, wk = sum(fv) / len(fv), sum(wv) / len(wv)
    rd, wr = fk * 1024 * 2, wk * 1024
    algo = ALGO[k] * NPIX
    out["kernels"][k] = {"FETCH_SIZE_KiB_avg_per_launch": fk, "WRITE_SIZE_KiB_avg_per_launch": wk, "launches_averaged": [len(fv), len(wv)],
                         "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr, "hbm_total_bytes": rd + wr, "algorithmic_bytes": algo,
                         "traffic_over_algorithmic": round((rd + wr) / algo, 4)}
json.dump(out, open(f"{dst}/pmc_hbm_traffic.json", "w"), indent=1)
open(os.path.join(os.path.dirname(dst.rstrip("/")), "LATEST"), "w").write(os.path.basename(dst.rstrip("/")) + "\n")
for k, v in out["kernels"].items():
    print(f"{k:26s} traffic {v['hbm_total_bytes']/1e6:9.1f} MB  = {v['traffic_over_algorithmic']:.4f} x algorithmic")
for pl in ("fused", "split"):
    p = f"{dst}/kernel_stats_{pl}.csv"
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            if kname(r["Name"]) in ALGO:
                print(pl, r["Name"][:60], "calls", r["Calls"], "avg ns", r["AverageNs"])
