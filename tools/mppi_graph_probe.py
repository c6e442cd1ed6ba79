"""Dev probe: is the K=1024 tick bound by the host's launch rate?  Times (a) plain enqueues, (b) a captured hipGraph of
100 ticks replayed, (c) the host-side cost of enqueueing alone."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
import bench
dev = torch.device("cuda", 0)
m = bench.make_mppi(1024, 0.5, 0)
T = m.steps
a, b = bench.synth_noise(T, 1024, dev, 1)
s = torch.cuda.Stream(dev)
sp = s.cuda_stream
X0 = bench.X0
for _ in range(200): m.enqueueDev(X0, a.data_ptr(), b.data_ptr(), sp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): m.enqueueDev(X0, a.data_ptr(), b.data_ptr(), sp)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"plain: host enqueue {t_enq / 2000 * 1e6:.2f} us/tick, end-to-end {t_all / 2000 * 1e6:.2f} us/tick")
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=s):
    for _ in range(100): m.enqueueDev(X0, a.data_ptr(), b.data_ptr(), sp)
torch.cuda.synchronize()
for _ in range(3): graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): graph.replay()
torch.cuda.synchronize()
print(f"graph of 100 ticks: {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us/tick")
print("controls", m.lastControls(sp))
