#!/bin/bash
# HBM traffic per launch by PMC, the way MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
# rocprofv3 passes (--kernel-trace --pmc only; no sys/hip/hsa tracing).  Run on the GPU box from the repo root:
#   tools/collect_pmc.sh <tag> <driver path relative to the repo root> [args...]
# writes gpurun_out/pmc_<tag>_{FETCH_SIZE,WRITE_SIZE}.csv (per-dispatch counter values) for tools/pmc_summary.py.
set -u
tag=$1; shift
root=$(pwd)
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${tag}_$ctr
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${tag}_$ctr -o run -- python "$root/$1" "${@:2}" > /tmp/pmc_${tag}_$ctr.log 2>&1
  f=$(find /tmp/pmc_${tag}_$ctr -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" "$root/gpurun_out/pmc_${tag}_$ctr.csv"; else echo "no counter csv for $ctr"; tail -5 /tmp/pmc_${tag}_$ctr.log; fi
done
