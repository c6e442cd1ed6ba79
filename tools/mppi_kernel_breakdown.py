"""Dev probe: per-kernel launch-to-launch intervals (HIP events) of the MPPI tick at mid ensemble sizes: python tools/mppi_kernel_breakdown.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.load_package()
import bench
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
for K, hor in ((4096, 1.0), (8192, 1.0), (16384, 1.0), (32768, 1.0), (8192, 0.5)):
    m = bench.make_mppi(K, hor, 0)
    a, b = bench.synth_noise(m.steps, K, dev, 1)
    ms = bench.kernel_profile(m, a, b, st, 200)
    print(K, m.steps, m.rollout_kernel[:36], "rollout %.1f us partials %.1f us combine %.1f us" % tuple(x * 1e3 for x in ms), "records/step", m.records_per_step, flush=True)
    m.close()
