"""A/B of the map update's form (0 = box counters, 1 = the beam-ordered kernel) and workgroup size on the bench
workload (BASELINE configs[2]): kernel time from HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
import torch
import bench_rbpf
from rtn_amd import capi
from rtn_amd.rbpf import ParticleFilter, default_params
steps, scans = bench_rbpf.workload(14)
for N in (1000, 4000):
    for form, nt in ((0, 0), (0, 512), (0, 1024), (1, 0)):
        pf = ParticleFilter(default_params(N=N, k=50, map_min=-10.0, map_max=10.0))
        pf.setSeed(1); pf.setTiming(True); pf.setOption(capi.RBPF_OPT_RAYCAST_ORDERED, form); pf.setOption(capi.RBPF_OPT_RAYCAST_THREADS, nt)
        acc, n = {}, 0
        for s, (prev, cur, t_icp, u) in enumerate(steps):
            pf.SLAM(scans[s], u, cur, prev, True, t_icp, None)
            if s >= 3:
                n += 1
                for k_, v in pf.kernelMs().items():
                    acc[k_] = acc.get(k_, 0.0) + v
        print(f"N={N} form={form} threads={nt}: raycast {acc['raycast'] / n * 1e3:.1f} us, propose {acc['propose'] / n * 1e3:.1f} us", flush=True)
        pf.close()
