"""Per-kernel HBM bytes per launch from the two CSVs tools/collect_pmc.sh leaves in gpurun_out/.

usage: python tools/pmc_summary.py <tag> [skip_first_n_launches]
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts a 128-B request as 64 B, so reads are doubled
(factor calibrated on mppi_partials' exactly-known read set, profiles/r01_traffic_pmc.json `_about`).
"""
import csv, json, re, sys
from collections import defaultdict

tag = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def short(name):
    name = re.sub(r"(\(anonymous namespace\)|tbnav_rk|tbnav_mk)::", "", name)
    return re.sub(r"^void ", "", name).split("(")[0]


def per_kernel(path, counter):
    vals = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            vals[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return {k: (sum(v[skip:]) / max(1, len(v[skip:])) if len(v) > skip else sum(v) / len(v), len(v)) for k, v in vals.items()}


rd = per_kernel(f"gpurun_out/pmc_{tag}_FETCH_SIZE.csv", "FETCH_SIZE")
wr = per_kernel(f"gpurun_out/pmc_{tag}_WRITE_SIZE.csv", "WRITE_SIZE")
out = {}
for k in sorted(set(rd) | set(wr)):
    if k.startswith("__amd") or "at::" in k:
        continue
    r, n = rd.get(k, (0.0, 0)); w, _ = wr.get(k, (0.0, 0))
    out[k] = {"launches": n, "fetch_size_kb_raw": round(r, 1), "write_size_kb_raw": round(w, 1),
              "read_bytes": int(r * 1024 * 2), "write_bytes": int(w * 1024), "hbm_bytes": int(r * 1024 * 2 + w * 1024)}
# the map update's launches one by one (launch order): the first scans, the plain ones and the one after a resampling differ
seq = {}
for path, counter in ((f"gpurun_out/pmc_{tag}_FETCH_SIZE.csv", "FETCH_SIZE"), (f"gpurun_out/pmc_{tag}_WRITE_SIZE.csv", "WRITE_SIZE")):
    with open(path) as f:
        rows = [r for r in csv.DictReader(f) if r.get("Counter_Name") == counter and "rbpf_raycast_box" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        seq.setdefault(short(r["Kernel_Name"]), {}).setdefault(counter, []).append(float(r["Counter_Value"]))
if seq and sum(len(v.get("FETCH_SIZE", [])) for v in seq.values()) <= 40:
    out["_map_update_launches_in_order"] = {k: {"read_bytes": " ".join(str(int(x * 2048)) for x in v.get("FETCH_SIZE", [])),
                                                 "write_bytes": " ".join(str(int(x * 1024)) for x in v.get("WRITE_SIZE", []))} for k, v in seq.items()}
print(json.dumps(out, indent=1))
