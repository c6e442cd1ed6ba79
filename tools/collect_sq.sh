#!/bin/bash
# SQ counters per kernel (instruction mix, busy / wait cycles) in ONE rocprofv3 pass (8 SQ slots; --kernel-trace --pmc
# only).  Run on the GPU box from the repo root:  tools/collect_sq.sh <tag> <driver relative to the repo root> [args...]
# writes gpurun_out/sq_<tag>.csv; summarise with tools/sq_summary.py <tag>.
set -u
tag=$1; shift
root=$(pwd)
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sq_${tag}
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d /tmp/sq_${tag} -o run -- python "$root/$1" "${@:2}" > /tmp/sq_${tag}.log 2>&1
f=$(find /tmp/sq_${tag} -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$root/gpurun_out/sq_${tag}.csv"; else echo "no counter csv"; tail -5 /tmp/sq_${tag}.log; fi
