#!/bin/bash
# Everything profiles/ is built from, in one go on the GPU box (run from the repo root):
#   1. rocprofv3 --kernel-trace --stats of `python bench.py` (the command the driver times)          -> gpurun_out/${R}_trace/
#   2. PMC passes (FETCH_SIZE, WRITE_SIZE separately; no other tracing) of the three workloads        -> gpurun_out/pmc_*.csv
#   3. SQ counter pass of the RBPF scan and the large MPPI tick                                       -> gpurun_out/sq_*.csv
set -u
R=${R:-r04}
root=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/${R}_trace
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d /tmp/${R}_trace -o run -- python "$root/bench.py" --steps 500 --warmup 50 --no-cpu-baseline > "$root/gpurun_out/${R}_bench_under_rocprof.json" 2> /tmp/${R}_trace.log
db=$(find /tmp/${R}_trace -name "*.db" | head -1)
if [ -n "$db" ]; then python "$root/profiles/summarize_rocpd.py" "$db" > "$root/gpurun_out/${R}_kernel_stats.md"; else echo "no rocpd db"; tail -5 /tmp/${R}_trace.log; find /tmp/${R}_trace | head; fi
cd "$root"
tools/collect_pmc.sh mppi_small tools/mppi_tick_driver.py 1024 0.5 300
tools/collect_pmc.sh mppi_small_rng tools/mppi_tick_driver.py 1024 0.5 300 rng
tools/collect_pmc.sh mppi_large tools/mppi_tick_driver.py 65536 1.0 40
tools/collect_pmc.sh rbpf tools/rbpf_driver.py 1000 14 dev
tools/collect_pmc.sh rbpf_plain tools/rbpf_driver.py 1000 14 plain
tools/collect_sq.sh rbpf tools/rbpf_driver.py 1000 14 dev
tools/collect_sq.sh mppi_large tools/mppi_tick_driver.py 65536 1.0 40
tools/collect_sq.sh mppi_small_rng tools/mppi_tick_driver.py 1024 0.5 300 rng
for t in mppi_small mppi_small_rng mppi_large rbpf rbpf_plain; do python tools/pmc_summary.py $t 3 > gpurun_out/pmc_summary_$t.json; done
for t in rbpf mppi_large mppi_small_rng; do python tools/sq_summary.py $t 3 > gpurun_out/sq_summary_$t.json; done
