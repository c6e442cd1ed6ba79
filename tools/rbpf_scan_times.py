"""Dev: per-scan wall time of the cfg3 RBPF run, scan by scan (first scans = nearly empty maps, i.e. lookups far from
any mapped obstacle), in the ICP-ok and the ICP-failed branch, for each distance-lookup mode."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
g.load_package()
import bench_rbpf, rbpf_cases as rc
from rtn_amd.rbpf import ParticleFilter, default_params
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_scans = 14
steps, poses = rc.trajectory(n_scans, inc=(0.07, 0.10, 0.05))
for icp_ok in (True, False):
    pf = ParticleFilter(default_params(N=N, k=50, map_min=-10.0, map_max=10.0))
    pf.setSeed(1)
    if "sm" in sys.argv[2:]:
        pf.setScanMatching(True)
    rng = np.random.default_rng(7)
    ts = []
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        walls = rc.ROOM_SURVEY
        if len(sys.argv) > 2 and sys.argv[2] == "explore" and s >= 6:   # two walls jump 1.2 m outwards: those beams now end ~24 cells from anything mapped
            walls = (walls[0] - 1.2, walls[1], walls[2], walls[3] + 1.2)
        scan = bench_rbpf._room_scan(poses[s], rng, walls)
        t0 = time.perf_counter()
        st = pf.SLAM(scan, u, cur, prev, icp_ok, t_icp, None)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("icp_ok" if icp_ok else "icp_failed", "query", " ".join(f"{t:.3f}" for t in ts), "ms; status", st.status)
    pf.close()
