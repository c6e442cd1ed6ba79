"""Soak of the in-kernel-noise hand-over (rbpf_propose's leading workgroup publishes the beam table, the others wait for it, bounded):
many synchronous scans at several ensemble sizes, every status checked; two filters interleaved on two streams' worth of handles.
python tools/rbpf_soak.py [scans]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
import numpy as np
import bench_rbpf
from rtn_amd import capi
from rtn_amd.rbpf import ParticleFilter, default_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
steps, scans = bench_rbpf.workload(40)
for N in (1, 37, 1000, 1025, 4000):
    pfs = [ParticleFilter(default_params(N=N, k=50, map_min=-10.0, map_max=10.0)) for _ in range(2)]
    for i, pf in enumerate(pfs):
        pf.setSeed(11 + i)
        pf.setOption(capi.RBPF_OPT_NOISE_IN_KERNEL, 1)   # (round 6: the option, no longer the default)
    t0 = time.perf_counter()
    m = n if N <= 1025 else n // 8
    for q in range(m):
        s = q % 40
        prev, cur, t_icp, u = steps[s]
        for pf in pfs:
            st = pf.SLAM(scans[s], u, cur, prev, True, t_icp, None)
            assert st.status == 0, (N, q, st.status)
        if s == 39:   # start over from fresh maps now and then (the trajectory repeats)
            pass
    dt = time.perf_counter() - t0
    a, b = pfs[0].particles(), pfs[1].particles()
    assert np.all(np.isfinite(a[0])) and np.all(np.isfinite(b[0]))
    print(f"N={N}: {2 * m} scans ok, {dt / (2 * m) * 1e6:.1f} us per scan ({pfs[0].lastKernelNames()[0]})", flush=True)
    for pf in pfs:
        pf.close()
print("soak ok")
