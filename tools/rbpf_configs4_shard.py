"""One GPU's shard of BASELINE configs[4] (100 000 particles / 8 GPUs = 12 500, 2000 x 2000 cells @ 0.05 m, 1080-beam scans,
k = 50): per-scan wall time and kernel times, with a forced resample in the run.  python tools/rbpf_configs4_shard.py [N] [k]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
g.load_package()
import bench_rbpf, rbpf_cases as rc
from rtn_amd.rbpf import ParticleFilter, default_params
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
bd = 1.0 / 3.0
n_scans = 10
pf = ParticleFilter(default_params(N=N, k=k, map_min=-50.0, map_max=50.0, beam_delta_deg=bd))
pf.setSeed(5); pf.setTiming(True)
if os.environ.get("RAYCAST_THREADS"):   # A-B runs: RAYCAST_THREADS=512, RAYCAST_ORDERED=1
    from rtn_amd import capi
    pf.setOption(capi.RBPF_OPT_RAYCAST_THREADS, int(os.environ["RAYCAST_THREADS"]))
if os.environ.get("RAYCAST_ADAPT"):     # RAYCAST_ADAPT=0: the LDS array sized for the scan's longest beam (round 3 before its last change)
    from rtn_amd import capi
    pf.setOption(capi.RBPF_OPT_RAYCAST_ADAPT, int(os.environ["RAYCAST_ADAPT"]))
if os.environ.get("RAYCAST_ORDERED"):
    from rtn_amd import capi
    pf.setOption(capi.RBPF_OPT_RAYCAST_ORDERED, int(os.environ["RAYCAST_ORDERED"]))
steps, poses = rc.trajectory(n_scans, inc=(0.05, 0.04, 0.03))
rng = np.random.default_rng(8)
scans = [bench_rbpf._room_scan(poses[s], rng, rc.ROOM_SURVEY, n_beams=1080, beam_delta_deg=bd) for s in range(n_scans)]
for s, (prev, cur, t_icp, u) in enumerate(steps):
    if s == 6:
        w = np.full(N, 1e-6); w[[7, N // 3, N // 2, N - 1]] = [0.4, 0.3, 0.2, 0.1]; w /= w.sum()
        pf.setParticles(w=w)
    t0 = time.perf_counter()
    st = pf.SLAM(scans[s], u, cur, prev, True, t_icp, None)
    dt = (time.perf_counter() - t0) * 1e3
    km = pf.kernelMs()
    cap, free, tb = pf.poolStats()
    print(f"scan {s}: {dt:7.3f} ms  valid beams {st.n_valid_beams}  resampled {st.resampled}  kernels " +
          ", ".join(f"{a} {b * 1e3:.0f} us" for a, b in km.items() if b) + f"  tiles in use {cap - free} ({(cap - free) * tb / 2**20:.0f} MiB)", flush=True)
pf.close()
