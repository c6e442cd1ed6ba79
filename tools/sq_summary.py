"""Per-kernel averages of the SQ counters tools/collect_sq.sh collected (gpurun_out/sq_<tag>.csv)."""
import csv, json, re, sys
from collections import defaultdict
tag = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 2
vals = defaultdict(lambda: defaultdict(list))
with open(f"gpurun_out/sq_{tag}.csv") as f:
    for row in csv.DictReader(f):
        name = re.sub(r"(\(anonymous namespace\)|tbnav_rk|tbnav_mk)::", "", row["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        vals[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, d in vals.items():
    if k.startswith("__amd") or "at::" in k:
        continue
    out[k] = {c: round(sum(v[skip:]) / max(1, len(v[skip:])), 1) if len(v) > skip else round(sum(v) / len(v), 1) for c, v in d.items()}
    out[k]["launches"] = len(next(iter(d.values())))
print(json.dumps(out, indent=1))
