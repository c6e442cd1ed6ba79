"""A-B of where the device noise is drawn (TBNAV_RBPF_OPT_NOISE_IN_KERNEL 1: inside rbpf_propose, whose leading workgroup carries the beam
table over; 0: rbpf_sample_normals first), with and without event timing: wall time per synchronous scan and the kernels' HIP-event
durations on the bench workload.  python tools/rbpf_noise_ab.py [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
import bench_rbpf
from rtn_amd import capi
from rtn_amd.rbpf import ParticleFilter, default_params
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps, scans = bench_rbpf.workload(40)
for rep in range(2):
    for in_kernel in (1, 0):
        for timing in (False, True):
            pf = ParticleFilter(default_params(N=N, k=50, map_min=-10.0, map_max=10.0))
            pf.setSeed(1); pf.setTiming(timing); pf.setOption(capi.RBPF_OPT_NOISE_IN_KERNEL, in_kernel)
            acc, n, wall = {}, 0, 0.0
            for s, (prev, cur, t_icp, u) in enumerate(steps):
                t0 = time.perf_counter()
                pf.SLAM(scans[s], u, cur, prev, True, t_icp, None)
                dt = time.perf_counter() - t0
                if s >= 8:
                    wall += dt; n += 1
                    if timing:
                        for k, v in pf.kernelMs().items(): acc[k] = acc.get(k, 0.0) + v
            km = ", ".join(f"{k} {v / n * 1e3:.1f}" for k, v in acc.items() if v) if timing else "-"
            print(f"N={N} in_kernel={in_kernel} timing={int(timing)}: {wall / n * 1e6:.1f} us per scan (wall); kernels [us]: {km}", flush=True)
            pf.close()
