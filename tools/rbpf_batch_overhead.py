"""What one tbnav_rbpf_slam_batch call costs on top of its scans: the bench workload (no resampling) replayed in calls of 1, 2, 3, 6,
12 and 24 scans.  python tools/rbpf_batch_overhead.py [N]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
import bench_rbpf
from rtn_amd.rbpf import ParticleFilter, default_params
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_scans = 56
steps, scans = bench_rbpf.workload(n_scans)
sc = np.stack(scans)
odom = np.array([steps[0][0]] + [st[1] for st in steps], dtype=np.float64)
u = np.array([st[3] for st in steps], dtype=np.float64)
t_icp = np.array([st[2] for st in steps], dtype=np.float64)
for rep in range(2):
    for L in (1, 2, 3, 6, 12, 24, 48):
        pf = ParticleFilter(default_params(N=N, k=50, map_min=-10.0, map_max=10.0)); pf.setSeed(3)
        pf.SLAMBatch(sc[:8], u[:8], odom[:9], t_icp[:8])
        t0 = time.perf_counter()
        for lo in range(8, n_scans, L):
            hi = min(lo + L, n_scans)
            pf.SLAMBatch(sc[lo:hi], u[lo:hi], odom[lo:hi + 1], t_icp[lo:hi])
        dt = time.perf_counter() - t0
        print(f"N={N}: calls of {L:2d} scans: {dt / (n_scans - 8) * 1e6:.1f} us per scan", flush=True)
        pf.close()
