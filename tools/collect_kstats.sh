#!/bin/bash
# Kernel statistics of ONE workload's driver: rocprofv3 --kernel-trace --stats (nothing else), summarised into the table bench_profiles.py
# parses.  Run on the GPU box from the repo root:   tools/collect_kstats.sh <tag> <driver path relative to the repo root> [args...]
# writes gpurun_out/kstats_<tag>.md
set -u
tag=$1; shift
root=$(pwd)
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kstats_${tag}
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/kstats_${tag} -o run -- python "$root/$1" "${@:2}" > /tmp/kstats_${tag}.log 2>&1
db=$(find /tmp/kstats_${tag} -name "*.db" | head -1)
if [ -n "$db" ]; then python "$root/profiles/summarize_rocpd.py" "$db" > "$root/gpurun_out/kstats_${tag}.md"; else echo "no rocpd db for $tag"; tail -5 /tmp/kstats_${tag}.log; fi
