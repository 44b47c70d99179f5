#!/usr/bin/env python3
"""Register / LDS figures of the kernels in a device assembly listing (development aid):
    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S csrc/conv_mfma.hip -o /tmp/conv.s
    python tools/kernel_regs.py /tmp/conv.s [substring ...]"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
md = t[t.rfind("amdhsa.kernels"):]
rows = []
for b in md.split("  - .agpr_count:")[1:]:
    g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", b).group(1))      # noqa: E731
    rows.append((re.search(r"\.name:\s+(\S+)", b).group(1), g("vgpr_count"), int(b.split("\n")[0].strip()), g("sgpr_count"),
                 g("vgpr_spill_count"), g("sgpr_spill_count"), g("group_segment_fixed_size")))
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.strip().split("\n")
for d, (n, v, a, s, sp, ssp, lds) in zip(names, rows):
    if len(sys.argv) < 3 or any(k in d for k in sys.argv[2:]):
        print(f"{d[:80]:80s} vgpr {v:3d} agpr {a:3d} sgpr {s:3d} vspill {sp} sspill {ssp}")
