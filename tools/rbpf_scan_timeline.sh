#!/bin/bash
# Device timeline of the kernels of the last scans of a cfg3 run (rocprofv3 --kernel-trace): start / end / gap to the previous
# kernel, in us.  Run on the GPU box from the repo root: tools/rbpf_scan_timeline.sh [N] [dev|batch]
root=$(pwd); N=${1:-1000}; mode=${2:-dev}   # mode: dev = one call per scan, batch = one tbnav_rbpf_slam_batch call
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o run -- python "$root/tools/rbpf_driver.py" $N 10 $mode > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-16:]
t0 = int(rows[0]["Start_Timestamp"]); prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"(\(anonymous namespace\)|tbnav_rk|tbnav_mk)::", "", r["Kernel_Name"]).split("(")[0][:40]
    print(f"{(s - t0) / 1e3:9.2f} {(e - t0) / 1e3:9.2f}  dur {(e - s) / 1e3:7.2f}  gap {((s - prev) / 1e3) if prev else 0:7.2f}  {name}")
    prev = e
PY
