#!/usr/bin/env python3
"""Turns the raw output of tools/profile_round.sh (merged back under gpurun_out/) into the committed summaries under
profiles/<name>/ and points profiles/LATEST at them.

    usage: make_profile_summary.py gpurun_out/<prof dir> profiles/<name> [--no-latest]

Per kernel and counter the whole-frame launches are selected by GRID SIZE (the largest grid the kernel was launched with:
bench.py also runs it on small check slices), and summarised by their MEDIAN; launches further than 2 % from the median
are counted as outliers and listed, never silently dropped or -- as the round-1 version of this tool did with its
"> 0.8 x max" rule -- allowed to evict every good sample.  Input layouts understood: round 2 (`stats/`, `pmc_<CTR>/`,
one default bench command whose second timed region runs the split pipeline) and round 1 (`stats_{fused,split}/`,
`pmc_<CTR>_{fused,split}/`).
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import statistics
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
src, dst = args[0], args[1]
set_latest = "--no-latest" not in sys.argv
os.makedirs(dst, exist_ok=True)
ALGO = {"k_fused_dma": 4, "k_fused_stream": 4, "k_gray_rgb_to_plane_w": 4, "k_sobel_dma2": 2, "k_sobel_dma": 2, "k_sobel_stream": 2, "k_copy_stream": 2,
        "k_gray_inplace_rgb_w": 6, "k_sobel_dma2_luma": 2, "k_gray_of_plane": 2}
NPIX = 16384 * 16384
NSIMD = 1024                      # 256 CUs x 4 SIMDs


def kname(s):
    m = re.search(r"(k_\w+)(<[^>]*>)?", s)
    if not m:
        return None
    # the plane Sobel with the luma applied on load (pass k of the multi-pass loop) is the same template with LUMA = 1, its last argument
    if m.group(1) == "k_sobel_dma2" and m.group(2) and m.group(2).replace(" ", "").endswith(",1>"):
        return "k_sobel_dma2_luma"
    return m.group(1)


def rows_of(pattern):
    for f in glob.glob(pattern):
        with open(f) as fh:
            yield from csv.DictReader(fh)


# ---- kernel statistics (rocprofv3 --kernel-trace --stats) ---------------------------------------------------------------
stats_out = {}
for tag, pat in (("", f"{src}/stats/*/*kernel_stats.csv"), ("_fused", f"{src}/stats_fused/*/*kernel_stats.csv"),
                 ("_split", f"{src}/stats_split/*/*kernel_stats.csv")):
    f = glob.glob(pat)
    if not f:
        continue
    shutil.copy(f[0], f"{dst}/kernel_stats{tag}.csv")
    for r in csv.DictReader(open(f[0])):
        k = kname(r["Name"])
        if k in ALGO:
            stats_out.setdefault(k, {})[tag or "default"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                                              "max_ns": float(r["MaxNs"])}
for f in glob.glob(f"{src}/bench*.json") + glob.glob(f"{src}/kernel_table.txt") + glob.glob(f"{src}/ab_*.txt") + glob.glob(f"{src}/skeleton*.txt"):
    if os.path.getsize(f):
        shutil.copy(f, dst)

# ---- counters -------------------------------------------------------------------------------------------------------------
acc = collections.defaultdict(lambda: collections.defaultdict(list))      # kernel -> counter -> [(grid, value)]
for r in rows_of(f"{src}/pmc_*/*/*counter_collection.csv"):
    k = kname(r["Kernel_Name"])
    if k in ALGO:
        acc[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))


def whole_frame(samples):
    gmax = max(g for g, _ in samples)
    v = sorted(x for g, x in samples if g == gmax)
    med = statistics.median(v)
    inl = [x for x in v if abs(x - med) <= 0.02 * abs(med)] or v
    outl = [x for x in v if x not in inl]
    return {"grid": gmax, "launches": len(v), "median": med, "mean_of_inliers": sum(inl) / len(inl), "min": v[0], "max": v[-1],
            "outliers_beyond_2pct": len(outl), "outlier_values": outl[:8]}


traffic = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of the bench command on one MI355X (tools/profile_round.sh); "
                   "16384x16384 frame; per kernel the launches with the largest grid (whole frame), summarised by the median; FETCH_SIZE doubled "
                   "(gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE as read", "kernels": {}}
for k, c in acc.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    f, w = whole_frame(c["FETCH_SIZE"]), whole_frame(c["WRITE_SIZE"])
    rd, wr = f["median"] * 1024 * 2, w["median"] * 1024
    algo = ALGO[k] * NPIX if k != "k_copy_stream" else 2 * 3 * NPIX          # the copy probe moves the 805 MB RGB frame: 1 R + 1 W of it
    traffic["kernels"][k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "launches_averaged": [f["launches"] - f["outliers_beyond_2pct"],
                                                                                           w["launches"] - w["outliers_beyond_2pct"]],
                             "hbm_read_bytes_corrected": rd, "hbm_write_bytes": wr, "hbm_total_bytes": rd + wr, "algorithmic_bytes": algo,
                             "traffic_over_algorithmic": round((rd + wr) / algo, 4)}
if traffic["kernels"]:
    json.dump(traffic, open(f"{dst}/pmc_hbm_traffic.json", "w"), indent=1)

# VALU issue: SQ_INSTS_VALU = wave-level VALU instructions; SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES / SQ_BUSY_CYCLES in quad-cycles
# (MI355X_MICROARCH.md, cycle-constants table); GRBM_GUI_ACTIVE = shader-clock cycles summed over the 8 XCDs.
valu = {"note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY "
                "GRBM_GUI_ACTIVE (one pass); medians over the whole-frame launches.  valu_instr_per_px = SQ_INSTS_VALU x 64 lanes / pixels "
                "(instructions one lane executes per pixel it owns); kernel_cycles = GRBM_GUI_ACTIVE / 8 XCDs; "
                "valu_issue_occupancy = 4 x SQ_ACTIVE_INST_VALU / (kernel_cycles x 1024 SIMDs); cycles_per_valu_instr = 4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU",
        "kernels": {}}
for k, c in acc.items():
    if "SQ_INSTS_VALU" not in c:
        continue
    m = {n: whole_frame(v) for n, v in c.items() if n.startswith("SQ_") or n.startswith("GRBM_")}
    med = {n: s["median"] for n, s in m.items()}
    e = {"launches": m["SQ_INSTS_VALU"]["launches"], "grid": m["SQ_INSTS_VALU"]["grid"], "medians": med}
    e["valu_instr_per_px"] = round(med["SQ_INSTS_VALU"] * 64 / NPIX, 2)
    if med.get("SQ_ACTIVE_INST_VALU") and med.get("SQ_INSTS_VALU"):
        e["cycles_per_valu_instr"] = round(4 * med["SQ_ACTIVE_INST_VALU"] / med["SQ_INSTS_VALU"], 2)
    if med.get("GRBM_GUI_ACTIVE"):
        cyc = med["GRBM_GUI_ACTIVE"] / 8
        e["kernel_cycles"] = round(cyc)
        if med.get("SQ_ACTIVE_INST_VALU"):
            e["valu_issue_occupancy"] = round(4 * med["SQ_ACTIVE_INST_VALU"] / (cyc * NSIMD), 3)
    valu["kernels"][k] = e
if valu["kernels"]:
    json.dump(valu, open(f"{dst}/pmc_valu_issue.json", "w"), indent=1)
if stats_out:
    json.dump(stats_out, open(f"{dst}/kernel_stats_summary.json", "w"), indent=1)

if set_latest:
    open(os.path.join(os.path.dirname(dst.rstrip("/")), "LATEST"), "w").write(os.path.basename(dst.rstrip("/")) + "\n")
for k, v in traffic["kernels"].items():
    print(f"{k:26s} traffic {v['hbm_total_bytes'] / 1e6:9.1f} MB = {v['traffic_over_algorithmic']:.4f} x algorithmic   launches {v['launches_averaged']}"
          f"  outliers F/W {v['FETCH_SIZE_KiB']['outliers_beyond_2pct']}/{v['WRITE_SIZE_KiB']['outliers_beyond_2pct']}")
for k, v in valu["kernels"].items():
    print(f"{k:26s} VALU instr/px {v['valu_instr_per_px']}  cycles/instr {v.get('cycles_per_valu_instr')}  issue occupancy {v.get('valu_issue_occupancy')}")
for k, v in stats_out.items():
    for tag, s in v.items():
        print(f"{k:26s} {tag:8s} calls {s['calls']:5d}  avg {s['avg_ns'] / 1e3:8.1f} us")
