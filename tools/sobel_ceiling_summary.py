#!/usr/bin/env python3
"""usage: sobel_ceiling_summary.py <gpurun_out/r4_sobel_ceiling> > profiles/r4_ab/sobel_ceiling_counters.md
Derived figures per kernel variant from the rocprofv3 counter passes of tools/profile_sobel_ceiling.sh (medians over the whole-frame
launches; counters are sums over their hardware instances: 256 TCP / TA, 128 TCC channels, 8 XCDs for GRBM_GUI_ACTIVE)."""
import json, subprocess, sys, os
here = os.path.dirname(os.path.abspath(__file__))
d = json.loads(subprocess.run([sys.executable, os.path.join(here, "pmc_by_variant.py"), sys.argv[1], "--json"], capture_output=True, text=True, check=True).stdout)
NAMES = {"k_sobel_dma2<3,1,1,0,0>": "two strips, ring depth 3 (ships)", "k_sobel_dma2<6,1,1,0,0>": "two strips, depth 6", "k_sobel_dma2<2,1,1,0,0>": "two strips, depth 2",
         "k_sobel_dma2<3,1,1,0,1>": "two strips, luma on load (pass k)", "k_sobel_dma<3,0,2,1,0,0,0>": "one strip, generic R=64", "k_sobel_dma<3,0,2,1,0,1,0>": "one strip, regular R=16 (3 waves/SIMD)",
         "k_sobel_dma<3,2,2,1,0,0,0>": "one strip, no arithmetic (skeleton)", "k_copy_stream": "linear copy",
         # tools/fused_ceiling.py (read bytes = 3 per pixel)
         "k_fused_dma<0,2,2,3,0,0>": "fused, ring depth 2 + pad (ships)", "k_fused_dma<0,3,2,3,0,0>": "fused, ring depth 3", "k_fused_dma<0,3,12,3,0,0>": "fused, no arithmetic (skeleton)",
         "k_gray_rgb_to_plane_w<0>": "linear RGB -> luma plane (same bytes)"}
READ_BYTES_PER_PX = 3 if any(k.startswith("k_fused_dma") for k in d) else 1
NPX = 16384 * 16384
cols = ["kernel time, us (GUI_ACTIVE / 8 / 2.4 GHz)", "L1->L2 read requests / (bytes / 128)", "outstanding L1 reads per CU", "L1 read latency, cycles", "EA read latency, cycles",
        "DRAM-credit stalls per EA read", "stalls per EA write", "TCP pending-stall, share of cycles", "TA busy", "L2 (TCC) busy", "UTCL1 misses", "VALU active / wave cycles",
        "wave parked (WAIT_ANY)", "issue stall (WAIT_INST_ANY)"]
print("| variant | " + " | ".join(cols) + " |")
print("|---|" + "---|" * len(cols))
for k, name in NAMES.items():
    v = d.get(k)
    if not v or "GRBM_GUI_ACTIVE" not in v:
        continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    row = [cyc / 2400, v["TCP_TCC_READ_REQ"] / (READ_BYTES_PER_PX * NPX / 128), v["TCP_TCC_READ_REQ_LATENCY"] / cyc / 256, v["TCP_TCC_READ_REQ_LATENCY"] / v["TCP_TCC_READ_REQ"],
           v["TCC_EA0_RDREQ_LEVEL"] / v["TCC_EA0_RDREQ"], v["TCC_EA0_RDREQ_DRAM_CREDIT_STALL"] / v["TCC_EA0_RDREQ"], v["TCC_EA0_WRREQ_STALL"] / v["TCC_EA0_WRREQ"],
           v["TCP_PENDING_STALL_CYCLES"] / 256 / cyc, v.get("TA_TA_BUSY", 0) / 256 / cyc, v["TCC_BUSY"] / 128 / cyc, v.get("TCP_UTCL1_TRANSLATION_MISS", 0),
           v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]]
    fmt = ["%.1f", "%.3f", "%.0f", "%.0f", "%.0f", "%.2f", "%.2f", "%.2f", "%.2f", "%.2f", "%.0f", "%.2f", "%.2f", "%.2f"]
    print(f"| {name} (`{k}`) | " + " | ".join(f % x for f, x in zip(fmt, row)) + " |")
