"""A/B of the rollout kernel at BASELINE configs[3]'s per-call size (K=65536, T=100) on one GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
bench.graft.load_package()
from rtn_amd import capi
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
sync = lambda: torch.cuda.synchronize(dev)
for K, H in ((65536, 1.0), (32768, 1.0), (65536, 0.6)):
    for name, opts in (("prefix", []), ("general", [(capi.MPPI_OPT_PREFIX_FORM, 0)])):
        m = bench.make_mppi(K, H, 0)
        for o, v in opts:
            m.setOption(o, v)
        a, b = bench.synth_noise(m.steps, K, dev, 99)
        el = bench.time_ticks(lambda: m.enqueueDev(bench.X0, a.data_ptr(), b.data_ptr(), stream), sync, 50, 10, lambda: None)
        ms = bench.kernel_profile(m, a, b, stream, 50)
        print(f"K={K} T={m.steps} {name} [{m.rollout_kernel}]: tick {el / 50 * 1e3:.4f} ms, rollout {ms[0] * 1e3:.1f} us ({24.0 * K * m.steps / (ms[0] * 1e-3) / 8e12 * 100:.1f} % of HBM peak), partials {ms[1] * 1e3:.1f}, combine {ms[2] * 1e3:.1f}", flush=True)
        m.close()
