"""Dev: duration of rbpf_normalize over a global weight vector of n entries (the sharded filter's per-scan exchange step),
HIP events round tbnav_rbpf_resample_global_dev's launch.  usage: python tools/normalize_time.py [n ...]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
import torch
from rtn_amd import capi
from rtn_amd.rbpf import ParticleFilter, default_params
pf = ParticleFilter(default_params(N=1, k=2))
for n in [int(a) for a in sys.argv[1:]] or [1000, 12500, 100000]:
    for name, w in (("no resample", np.random.default_rng(1).random(n) + 0.5), ("resample", np.where(np.arange(n) % 97 == 5, 1.0, 1e-6 * np.random.default_rng(2).random(n)))):
        dw = torch.from_numpy(w).cuda()
        parents = np.empty(n, dtype=np.int32); st = capi.RbpfStats()
        ts = []
        for _ in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            capi.check(pf._L.tbnav_rbpf_resample_global_dev(pf._h, dw.data_ptr(), n, 0, C.c_double(0.3), parents.ctypes.data, C.byref(st)), "x")
            ts.append(time.perf_counter() - t0)
        print(f"n={n:7d} {name:12s} resampled={st.resampled} wall per call {min(ts) * 1e6:8.1f} us (kernel + one sync + copies)", flush=True)
