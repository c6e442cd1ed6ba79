"""(no GPU needed) Global / scratch memory instructions, the waits on them and the barriers of one kernel, mapped to source lines.
usage: isa_memops.py <file.hip> <mangled-name regex> [first source line] [extra hipcc flags ...]
What showed that every pair of rbpf_raycast_box's final pass waited for the previous pair's store (vmcnt counts stores on gfx950)."""
import os, re, subprocess, sys, tempfile
src, key = sys.argv[1], sys.argv[2]
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), "isa_memops.s")
contract = "-ffp-contract=off" if "rbpf" in src else "-ffp-contract=fast-honor-pragmas"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", f"-I{root}/include",
                f"-I{os.path.dirname(os.path.abspath(src))}", contract, "-gline-tables-only", "-S", "--cuda-device-only", src, "-o", out]
               + sys.argv[4:], check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^\S*" + key + r"\S*:", l)][0]
cur = None
for i in range(start, len(lines)):
    l = lines[i]
    m = re.match(r"\s*\.loc\s+\d+\s+(\d+)\s+(\d+)", l)
    if m:
        cur = int(m.group(1)); continue
    if ".Lfunc_end" in l: break
    if re.search(r"global_load|global_store|global_atomic|s_barrier|s_waitcnt.*vmcnt|scratch_|buffer_", l) and (cur or 0) >= first:
        print(cur, l.strip().split(";")[0])
