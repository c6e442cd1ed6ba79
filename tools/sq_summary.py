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
seq = defaultdict(lambda: defaultdict(list))   # the map update's launches one by one (launch order)
with open(f"gpurun_out/sq_{tag}.csv") as f:
    rows = [r for r in csv.DictReader(f) if "rbpf_raycast_box" in r["Kernel_Name"] and r["Counter_Name"] in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
for r in rows:
    name = re.sub(r"(\(anonymous namespace\)|tbnav_rk|tbnav_mk)::", "", r["Kernel_Name"])
    seq[re.sub(r"^void ", "", name).split("(")[0]][r["Counter_Name"]].append(int(float(r["Counter_Value"])))
if seq and sum(len(next(iter(v.values()))) for v in seq.values()) <= 40:
    out["_map_update_launches_in_order"] = {k: {c: " ".join(map(str, x)) for c, x in v.items()} for k, v in seq.items()}
print(json.dumps(out, indent=1))
