#!/usr/bin/env python
"""Basic-block census of one kernel in a hipcc -S listing: per block the number of VALU / SALU / LDS / VMEM instructions and
s_waitcnt's, loop membership as the listing marks it.  python tools/isa_blocks.py file.s mangled-name-substring [min_instrs]"""
import collections
import re
import sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 0
m = re.search(r"^(\S*" + re.escape(key) + r"\S*):", s, re.M)
i = m.start()
j = s.index(".Lfunc_end", i)
blocks, cur = [], ["entry", []]
blocks.append(cur)
for ln in s[i:j].splitlines()[1:]:
    if re.match(r"^\.LBB\d+_\d+:", ln):
        cur = [ln.strip(), []]
        blocks.append(cur)
    elif ln.strip() and not ln.strip().startswith(";") and not ln.strip().startswith("."):
        cur[1].append(ln.strip())
tot = collections.Counter()
for lab, ins in blocks:
    c = collections.Counter()
    for x in ins:
        op = x.split()[0]
        kind = ("wait" if op.startswith("s_waitcnt") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
                "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other")
        c[kind] += 1
        if op.startswith("v_accvgpr"): c["acc"] += 1
        if op in ("v_readlane_b32", "v_writelane_b32"): c["lane"] += 1
    tot.update(c)
    if len(ins) >= minn:
        print(f"{lab[:70]:70s} {len(ins):5d} {dict(c)}")
print("total", dict(tot))
