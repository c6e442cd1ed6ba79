"""Dev driver: run N MPPI ticks at (K, horizon) for profiling under rocprofv3."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
import bench
K = int(sys.argv[1]); hor = float(sys.argv[2]); n = int(sys.argv[3])
dev = torch.device("cuda", 0)
m = bench.make_mppi(K, hor, 0)
a, b = bench.synth_noise(m.steps, K, dev, 1)
st = torch.cuda.current_stream(dev).cuda_stream
rng = len(sys.argv) > 4 and sys.argv[4] == "rng"   # production tick: perturbations drawn inside the kernel
for i in range(n):
    if rng:
        m.enqueueRng(bench.X0, 42, i, st)
    else:
        m.enqueueDev(bench.X0, a.data_ptr(), b.data_ptr(), st)
print(m.lastControls(st))
