"""Reference-field mode on the bench workload (configs[2]): ms per scan for a few TBNAV_RBPF_OPT_REF_REACH values, with the lazy
brushfire's counters (tbnav_rbpf_reference_field_stats).  python tools/ref_mode_probe.py [n_scans] [reach ...]
TBNAV_PROBE_ROOM=survey: SURVEY 8-d's room and trajectory instead of the bench's."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
import bench_rbpf as br  # noqa: E402
from rtn_amd import capi  # noqa: E402
from rtn_amd.rbpf import ParticleFilter, default_params  # noqa: E402

N = int(os.environ.get("TBNAV_PROBE_N", "1000"))
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reaches = [int(x) for x in sys.argv[2:]] or [6, 3, 0]
if os.environ.get("TBNAV_PROBE_ROOM") == "survey":
    import rbpf_cases as rc
    steps, poses = rc.trajectory(n_scans, inc=rc.TRAJ_SURVEY)
    rng = np.random.default_rng(7)
    scans = [br._room_scan(poses[s], rng, rc.ROOM_SURVEY) for s in range(n_scans)]
else:
    steps, scans = br.workload(n_scans)
for reach in reaches:
    pf = ParticleFilter(default_params(N=N, k=50, map_min=-10.0, map_max=10.0, device=0), df_mode="reference")
    pf.setOption(capi.RBPF_OPT_REF_REACH, reach)
    if os.environ.get("TBNAV_PROBE_THREADS"):
        pf.setOption(capi.RBPF_OPT_HOST_THREADS, int(os.environ["TBNAV_PROBE_THREADS"]))
    pf.setSeed(2026)
    t, per = 0.0, []
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        t0 = time.perf_counter()
        st = pf.SLAM(scans[s], u, cur, prev, True, t_icp, None)
        dt = time.perf_counter() - t0
        per.append(round(dt * 1e3, 3))
        if s >= 1:
            t += dt
    out = {"threads": os.environ.get("TBNAV_PROBE_THREADS"), "reach": reach, "ms_per_scan": round(t / (n_scans - 1) * 1e3, 3), "N": N, "updates_per_s": round(N * (n_scans - 1) / t, 1),
           "stats": pf.referenceFieldStats(), "counts": pf.referenceFieldCounts(), "per_scan_ms": per}
    print(json.dumps(out), flush=True)
    pf.close()
