"""Timeline of the map update that FOLLOWS a resampling (the launch that makes the written tiles of a shared map private): the
weights are skewed before scan 8, scan 8 resamples, scan 9 is the last launch before the handle is closed — a library built with
EXTRA='-DTBNAV_PHASE_PROF -DTBNAV_TRACE_ONLY' prints the wave stamps of two of its workgroups and the residency of all of them.
TBNAV_DEV_LIB=<that build>.  argv[1]: scans to run after the resampling one (default 1: the launch traced is the first after it;
2: the second, a plain scan, for comparison)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
import bench_rbpf
from rtn_amd import capi
from rtn_amd.rbpf import ParticleFilter, default_params
if os.environ.get("TBNAV_DEV_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["TBNAV_DEV_LIB"])
after = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = 1000
steps, scans = bench_rbpf.workload(9 + after)
pf = ParticleFilter(default_params(N=N, k=50, map_min=-10.0, map_max=10.0))
pf.setSeed(2026)
if os.environ.get("TBNAV_COW_TIMING"):   # (event timing instead of the trace: an ordinary build)
    pf.setTiming(True)
for s, (prev, cur, t_icp, u) in enumerate(steps):
    if s == 8:
        bench_rbpf._skew(pf, N)
    st = pf.SLAM(scans[s], u, cur, prev, True, t_icp, None)
    print(s, int(st.resampled), pf.lastKernelNames()[1], round(pf.kernelMs().get("raycast", 0.0) * 1e3, 1), flush=True)
pf.close()
