#!/usr/bin/env python3
"""VGPRs / spills / scratch of every kernel in a device-only assembly listing:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S x.hip -o x.s && tools/kernel_regs.py x.s [substring]"""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
    name = g("name")
    if pat in name:
        print(re.sub(r"^_ZN6speech12_GLOBAL__N_1\d+", "", name), "vgpr", g("vgpr_count"), "spill", g("vgpr_spill_count"), "scratch", g("private_segment_fixed_size"), "sgpr_spill", g("sgpr_spill_count"))
