#!/bin/bash
# Any set of counters (at most 8 SQ slots per pass) per kernel: tools/collect_counters.sh <tag> "<CTR1 CTR2 ...>" <driver> [args...]
# -> gpurun_out/sq_<tag>.csv (summarise with tools/sq_summary.py <tag>).  --kernel-trace --pmc only.
set -u
tag=$1; ctrs=$2; shift 2
root=$(pwd)
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sq_${tag}
timeout -k 5 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/sq_${tag} -o run -- python "$root/$1" "${@:2}" > /tmp/sq_${tag}.log 2>&1
f=$(find /tmp/sq_${tag} -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$root/gpurun_out/sq_${tag}.csv"; else echo "no counter csv"; tail -5 /tmp/sq_${tag}.log; fi
