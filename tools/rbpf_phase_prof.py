"""Per-phase timing of the proposal and raycast kernels (build csrc with EXTRA=-DTBNAV_PHASE_PROF first; the stamps
add barriers and atomics, so the numbers compare phases, they are not the bench)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
import torch
import bench_rbpf
from rtn_amd import capi
from rtn_amd.rbpf import ParticleFilter, default_params
steps, scans = bench_rbpf.workload(12)
if os.environ.get("TBNAV_DEV_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["TBNAV_DEV_LIB"])
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
pf = ParticleFilter(default_params(N=N, k=50, map_min=-10.0, map_max=10.0))
pf.setSeed(1); pf.setOption(capi.RBPF_OPT_RAYCAST_THREADS, nt)
if len(sys.argv) > 3: pf.setOption(capi.RBPF_OPT_NOISE_IN_KERNEL, int(sys.argv[3]))   # 0: rbpf_propose<., false> (stored normals)
for s, (prev, cur, t_icp, u) in enumerate(steps):
    pf.SLAM(scans[s], u, cur, prev, True, t_icp, None)
pf.close()
