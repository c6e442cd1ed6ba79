"""Is there anything to gain from running the proposal kernel of one half of the particles beside the map update of the other half?
Upper-bound experiment: TWO independent filters of N/2 particles, each replaying the bench workload through tbnav_rbpf_slam_batch from
its own host thread (own stream: their kernels overlap however the hardware lets them), against ONE filter of N.
python tools/rbpf_two_halves.py [N] [scans]"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
import bench_rbpf
from rtn_amd.rbpf import ParticleFilter, default_params
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 60
steps, scans = bench_rbpf.workload(n_scans)
sc = np.stack(scans)
odom = np.array([steps[0][0]] + [st[1] for st in steps], dtype=np.float64)
u = np.array([st[3] for st in steps], dtype=np.float64)
t_icp = np.array([st[2] for st in steps], dtype=np.float64)
def make(n, seed):
    pf = ParticleFilter(default_params(N=n, k=50, map_min=-10.0, map_max=10.0)); pf.setSeed(seed)
    pf.SLAMBatch(sc[:8], u[:8], odom[:9], t_icp[:8])
    return pf
def run(pf, out, i):
    t0 = time.perf_counter()
    pf.SLAMBatch(sc[8:], u[8:], odom[8:], t_icp[8:])
    out[i] = time.perf_counter() - t0
for rep in range(3):
    pf = make(N, 1); out = [0.0]; run(pf, out, 0); pf.close()
    print(f"one filter of {N}: {out[0] / (n_scans - 8) * 1e6:.1f} us per scan", flush=True)
    pf = make(N // 2, 1); out = [0.0]; run(pf, out, 0); pf.close()
    print(f"one filter of {N // 2}: {out[0] / (n_scans - 8) * 1e6:.1f} us per scan", flush=True)
    pfs = [make(N // 2, 1), make(N // 2, 2)]; out = [0.0, 0.0]
    th = [threading.Thread(target=run, args=(pfs[i], out, i)) for i in range(2)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    wall = time.perf_counter() - t0
    print(f"two filters of {N // 2} side by side: {wall / (n_scans - 8) * 1e6:.1f} us per scan of both (each thread {out[0] / (n_scans - 8) * 1e6:.1f} / {out[1] / (n_scans - 8) * 1e6:.1f})", flush=True)
    for p in pfs: p.close()
