"""Dev driver: a few RBPF scans of ONE workload for profiling under rocprofv3.
  python tools/rbpf_driver.py N n_scans [host|dev|plain|batch] [bench|survey|cfg4]
mode   host: reference-order normals from the host; dev: device noise with the bench's two forced resamplings; plain: device noise, no
       forced resampling (every map update a plain scan: no tile clones); batch: the run as one tbnav_rbpf_slam_batch call
room   bench: bench_rbpf.py's headline room (360 valid beams, 400^2); survey: SURVEY 8-d's 6 x 5 m room (246 valid beams, 400^2);
       cfg4: the per-GPU shard shape of BASELINE configs[4] (2000^2 cells, 1080-beam scans, SURVEY room)"""
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
mode = sys.argv[3] if len(sys.argv) > 3 else "host"
room = sys.argv[4] if len(sys.argv) > 4 else "bench"
dev_noise = mode in ("dev", "plain")   # standard normals drawn on the device (bench mode)
no_resample = mode == "plain"
if room == "cfg4":
    bd = 1.0 / 3.0
    pf = ParticleFilter(default_params(N=N, k=50, map_min=-50.0, map_max=50.0, beam_delta_deg=bd), pool_bytes=16 << 30)
    steps, poses = rc.trajectory(n_scans, inc=(0.05, 0.04, 0.03))
    rng = np.random.default_rng(7)
    scans = [bench_rbpf._room_scan(poses[s], rng, rc.ROOM_SURVEY, n_beams=1080, beam_delta_deg=bd) for s in range(n_scans)]
elif room == "survey":
    pf = ParticleFilter(default_params(N=N, k=50, map_min=-10.0, map_max=10.0))
    steps, poses = rc.trajectory(n_scans, inc=rc.TRAJ_SURVEY)
    rng = np.random.default_rng(7)
    scans = [bench_rbpf._room_scan(poses[s], rng, rc.ROOM_SURVEY) for s in range(n_scans)]
else:
    pf = ParticleFilter(default_params(N=N, k=50, map_min=-10.0, map_max=10.0))
    steps, scans = bench_rbpf.workload(n_scans)   # the bench's room: all 360 beams valid
pf.setSeed(2026)
if mode == "batch":   # the whole run as ONE tbnav_rbpf_slam_batch call (two scans in the stream)
    odom = np.array([steps[0][0]] + [st[1] for st in steps]); u_all = np.array([st[3] for st in steps]); t_all = np.array([st[2] for st in steps])
    out = pf.SLAMBatch(np.stack(scans[:n_scans]), u_all, odom, t_all)
    print(out[-1].neff)
    sys.exit(0)
for s, (prev, cur, t_icp, u) in enumerate(steps):
    scan = scans[s]
    if s in bench_rbpf.RESAMPLE_AT and not no_resample:
        bench_rbpf._skew(pf, N)
    st = pf.SLAM(scan, u, cur, prev, True, t_icp, None if dev_noise else np.random.default_rng(100 + s).standard_normal(pf.numNormals(True)))
print(st.neff, pf.lastKernelNames())
