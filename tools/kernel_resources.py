#!/usr/bin/env python
"""Compile one kernel file for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
VGPRs, AGPRs, SGPRs, scratch bytes per lane, LDS bytes, occupancy (waves per SIMD).  Runs without a GPU.
  python tools/kernel_resources.py mppi|rbpf [name-filter] [extra hipcc flags...]"""
import re
import subprocess
import sys

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ros-turtlebot-navigation_amd", "csrc")
which = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
contract = "-ffp-contract=fast-honor-pragmas" if which.startswith("mppi") else "-ffp-contract=off"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", f"-I{ROOT}/include", f"-I{CSRC}",
       contract, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, which + ".hip"), "-o", "/tmp/_kr.o"] + extra
err = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
demangle = subprocess.run(["c++filt"], input=err, stdout=subprocess.PIPE, text=True).stdout
cur = None
rows = {}
for line in demangle.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+(?: \[[^\]]*\])?): (.*) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = re.sub(r"(\(anonymous namespace\)|tbnav_rk)::", "", v)
        cur = re.sub(r"^void ", "", cur).split("(")[0]
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
print(f"{'kernel':60s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
for name, r in rows.items():
    if flt and flt not in name:
        continue
    print(f"{name[:60]:60s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>8s} {r.get('LDS Size [bytes/block]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>4s}")
