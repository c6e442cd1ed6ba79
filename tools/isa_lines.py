"""(no GPU needed) Static instruction count per source line of one kernel (VALU / SALU / LDS / VMEM), from a -gline-tables-only listing.
usage: isa_lines.py <file.hip> <mangled-name regex> [first line] [last line]"""
import os, re, subprocess, sys, tempfile
from collections import defaultdict
src, key = sys.argv[1], sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 10**9
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), "isa_lines.s")
contract = "-ffp-contract=off" if "rbpf" in src else "-ffp-contract=fast-honor-pragmas"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", f"-I{root}/include",
                f"-I{os.path.dirname(os.path.abspath(src))}", contract, "-gline-tables-only", "-S", "--cuda-device-only", src, "-o", out],
               check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^\S*" + key + r"\S*:", l)][0]
cnt = defaultdict(lambda: [0, 0, 0, 0])
cur = 0
for i in range(start + 1, len(lines)):
    l = lines[i]
    m = re.match(r"\s*\.loc\s+\d+\s+(\d+)\s+(\d+)", l)
    if m:
        cur = int(m.group(1)); continue
    if ".Lfunc_end" in l: break
    t = l.strip()
    if not t or t.startswith((".", ";")) or t.endswith(":"): continue
    op = t.split()[0]
    k = 0 if op.startswith("v_") else 1 if op.startswith("s_") else 2 if op.startswith("ds_") else 3
    cnt[cur][k] += 1
tot = [0, 0, 0, 0]
text = open(src).read().split("\n")
for ln in sorted(cnt):
    if lo <= ln <= hi:
        c = cnt[ln]
        for k in range(4): tot[k] += c[k]
        print(f"{ln:5d} V{c[0]:4d} S{c[1]:4d} L{c[2]:3d} M{c[3]:3d} | {text[ln - 1].strip()[:110]}")
print("total (static) VALU %d SALU %d LDS %d VMEM %d" % tuple(tot))
