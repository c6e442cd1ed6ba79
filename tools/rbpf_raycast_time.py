"""Kernel time of the map update on the bench workload (BASELINE configs[2]) from HIP events, three filters per particle count:
python tools/rbpf_raycast_time.py [N ...]
(TBNAV_DEV_LIB=<path to another build of libtbnav_hip.so> for A/B runs of two builds in one gpurun call — read here, not by the package)"""
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
steps, scans = bench_rbpf.workload(16)
for N in [int(a) for a in sys.argv[1:]] or [1000, 4000]:
    for rep in range(3):
        pf = ParticleFilter(default_params(N=N, k=50, map_min=-10.0, map_max=10.0))
        pf.setSeed(1); pf.setTiming(True)
        acc, n = {}, 0
        for s, (prev, cur, t_icp, u) in enumerate(steps):
            pf.SLAM(scans[s], u, cur, prev, True, t_icp, None)
            if s >= 5:
                for k, v in pf.kernelMs().items(): acc[k] = acc.get(k, 0.0) + v
                n += 1
        print(f"N={N}: raycast {acc['raycast'] / n * 1e3:.1f} us, propose {acc['propose'] / n * 1e3:.1f} us  ({pf.lastKernelNames()[1]})", flush=True)
        pf.close()
