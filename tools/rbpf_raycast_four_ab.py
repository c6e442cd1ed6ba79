"""A-B of rbpf_raycast_box's residency on the bench workload: TBNAV_RBPF_OPT_RAYCAST_ADAPT 1 (four 512-thread workgroups per CU when the
boxes fit) against 2 (at most three) and 3 (four-event slots wherever four fit) — kernel time from HIP events and which instantiation ran."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
import torch
import bench_rbpf
from rtn_amd import capi
from rtn_amd.rbpf import ParticleFilter, default_params
steps, scans = bench_rbpf.workload(16)
if os.environ.get("WALLS"):   # e.g. WALLS=-2.0,2.0,-1.8,1.8: a room whose box fits the four-per-CU form (the bench room's does not, by 2 KB)
    import numpy as np
    walls = tuple(float(v) for v in os.environ["WALLS"].split(","))
    rc = bench_rbpf._world()
    _, poses = rc.trajectory(16, inc=bench_rbpf.TRAJ_INC)
    rng = np.random.default_rng(7)
    scans = [bench_rbpf._room_scan(poses[s], rng, walls) for s in range(16)]
for N in [int(a) for a in sys.argv[1:]] or [1000, 2000, 4000]:
    for adapt, cell16 in ((2, 0), (1, 0), (3, 0), (2, 0), (1, 0), (3, 0), (1, 1), (2, 2)):
        pf = ParticleFilter(default_params(N=N, k=50, map_min=-10.0, map_max=10.0))
        pf.setSeed(1); pf.setTiming(True); pf.setOption(capi.RBPF_OPT_RAYCAST_ADAPT, adapt); pf.setOption(capi.RBPF_OPT_RAYCAST_CELL16, cell16)
        acc, n = 0.0, 0
        for s, (prev, cur, t_icp, u) in enumerate(steps):
            pf.SLAM(scans[s], u, cur, prev, True, t_icp, None)
            if s >= 5:
                acc += pf.kernelMs()["raycast"]; n += 1
        print(f"N={N} adapt={adapt} cell16={cell16}: raycast {acc / n * 1e3:.1f} us  ({pf.lastKernelNames()[1]}; box need / array cells {pf.raycastBoxCells()})", flush=True)
        pf.close()
