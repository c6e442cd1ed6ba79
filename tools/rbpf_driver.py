"""Dev driver: a few cfg3 RBPF scans for profiling under rocprofv3."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
g.load_package()
import bench_rbpf, rbpf_cases as rc
from rtn_amd.rbpf import ParticleFilter, default_params
if os.environ.get("TBNAV_DEV_LIB"):   # (A/B runs of two builds of the library in one gpurun call)
    from rtn_amd import capi
    capi.LIB_PATH = os.path.abspath(os.environ["TBNAV_DEV_LIB"])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev_noise = len(sys.argv) > 3 and sys.argv[3] in ("dev", "plain")   # standard normals drawn on the device (bench mode)
no_resample = len(sys.argv) > 3 and sys.argv[3] == "plain"         # no forced resample: every map update is a plain scan (no tile clones)
pf = ParticleFilter(default_params(N=N, k=50, map_min=-10.0, map_max=10.0))
steps, scans = bench_rbpf.workload(n_scans)   # the bench's room: all 360 beams valid
if len(sys.argv) > 3 and sys.argv[3] == "batch":   # the whole run as ONE tbnav_rbpf_slam_batch call (two scans in the stream)
    odom = np.array([steps[0][0]] + [st[1] for st in steps]); u_all = np.array([st[3] for st in steps]); t_all = np.array([st[2] for st in steps])
    out = pf.SLAMBatch(np.stack(scans[:n_scans]), u_all, odom, t_all)
    print(out[-1].neff)
    sys.exit(0)
for s, (prev, cur, t_icp, u) in enumerate(steps):
    scan = scans[s]
    if s in bench_rbpf.RESAMPLE_AT and not no_resample:
        bench_rbpf._skew(pf, N)
    st = pf.SLAM(scan, u, cur, prev, True, t_icp, None if dev_noise else np.random.default_rng(100 + s).standard_normal(pf.numNormals(True)))
print(st.neff)
