#!/bin/bash
# Instruction-cache counters per kernel in ONE rocprofv3 pass (--kernel-trace --pmc only).  Run on the GPU box from the repo root:
#   tools/collect_icache.sh <tag> <driver relative to the repo root> [args...]   -> gpurun_out/icache_<tag>.csv, and a summary on stdout
set -u
tag=$1; shift
root=$(pwd)
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ic_${tag}
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY \
  --output-format csv -d /tmp/ic_${tag} -o run -- python "$root/$1" "${@:2}" > /tmp/ic_${tag}.log 2>&1
f=$(find /tmp/ic_${tag} -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$root/gpurun_out/icache_${tag}.csv"; else echo "no counter csv"; tail -5 /tmp/ic_${tag}.log; exit 1; fi
python - "$root/gpurun_out/icache_${tag}.csv" <<'PY'
import csv, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("tbnav_rk::", "").replace("tbnav_mk::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r.get("Dispatch_Id"))
    if key not in seen: seen.add(key); n[k] += 1
for k, v in acc.items():
    if n[k] < 3: continue
    print(k, "launches", n[k], {c: round(x / n[k], 1) for c, x in v.items()})
PY
