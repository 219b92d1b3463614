#!/usr/bin/env python3
"""Register / scratch figures of the LDS-DMA kernels from `make isa` (build_isa/*.s): one line per instantiation.  Developer tool."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for f in ("sobel", "fused", "conv3x3"):
    t = open(os.path.join(ROOT, "build_isa", f + ".s")).read()
    for blk in t.split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
        name = g("name")
        if "_dma" not in name:
            continue
        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        d = re.sub(r"\(.*", "", d).replace("void (anonymous namespace)::", "")
        print(f"{f:8s} {d:48s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>3s} sgpr {g('sgpr_count'):>3s} scratch {g('private_segment_fixed_size')} lds {g('group_segment_fixed_size')}")
