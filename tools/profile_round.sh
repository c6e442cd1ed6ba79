#!/bin/bash
# Everything profiles/ is built from, in one go on the GPU box (run from the repo root):
#   1. rocprofv3 --kernel-trace --stats of `python bench.py` (the command the driver times)          -> gpurun_out/${R}_trace/
#   2. per workload (one driver each): a --kernel-trace --stats pass (its OWN table: the bench run mixes the legs' launches),
#      PMC passes (FETCH_SIZE, WRITE_SIZE separately; no other tracing) and an SQ counter pass        -> gpurun_out/{kstats,pmc,sq}_*
# Workloads (round 5: one per BASELINE shape a roofline object is quoted for):
#   mppi_small[_rng]  K=1024,T=50 (configs[1])            mppi_mid_rng  K=8192,T=100 (configs[3] / 8)     mppi_large  K=65536,T=100
#   rbpf / rbpf_plain N=1000 x 400^2 bench room (configs[2])   rbpf_survey  the same on SURVEY 8-d's room   rbpf_cfg4  N=12500 x 2000^2 x 1080 beams (configs[4] / 8)
set -u
R=${R:-r06}
root=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/${R}_trace
timeout -k 5 900 rocprofv3 --kernel-trace --stats -d /tmp/${R}_trace -o run -- python "$root/bench.py" --steps 500 --warmup 50 --no-cpu-baseline > "$root/gpurun_out/${R}_bench_under_rocprof.json" 2> /tmp/${R}_trace.log
db=$(find /tmp/${R}_trace -name "*.db" | head -1)
if [ -n "$db" ]; then python "$root/profiles/summarize_rocpd.py" "$db" > "$root/gpurun_out/${R}_kernel_stats.md"; else echo "no rocpd db"; tail -5 /tmp/${R}_trace.log; find /tmp/${R}_trace | head; fi
cd "$root"
run_all() {  # tag, driver, args...: kernel stats, PMC, SQ of one workload
  tools/collect_kstats.sh "$@"
  tools/collect_pmc.sh "$@"
  tools/collect_sq.sh "$@"
}
run_all mppi_small tools/mppi_tick_driver.py 1024 0.5 300
run_all mppi_small_rng tools/mppi_tick_driver.py 1024 0.5 300 rng
run_all mppi_mid_rng tools/mppi_tick_driver.py 8192 1.0 200 rng
run_all mppi_large tools/mppi_tick_driver.py 65536 1.0 40
run_all rbpf tools/rbpf_driver.py 1000 14 dev
run_all rbpf_plain tools/rbpf_driver.py 1000 14 plain
run_all rbpf_survey tools/rbpf_driver.py 1000 14 plain survey
run_all rbpf_cfg4 tools/rbpf_driver.py 12500 8 plain cfg4
for t in mppi_small mppi_small_rng mppi_mid_rng mppi_large rbpf rbpf_plain rbpf_survey rbpf_cfg4; do
  python tools/pmc_summary.py $t 3 > gpurun_out/pmc_summary_$t.json
  python tools/sq_summary.py $t 3 > gpurun_out/sq_summary_$t.json
done
