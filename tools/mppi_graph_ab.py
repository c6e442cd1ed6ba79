"""Dev probe: tbnav_mppi_enqueue_rng_batch with hipGraph replay of 100-tick chunks against plain launches (TBNAV_MPPI_OPT_BATCH_GRAPH)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.load_package()
import bench
from rtn_amd import capi
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(dev))
st = torch.cuda.current_stream(dev).cuda_stream
for K, hor in ((1024, 0.5), (2048, 0.5), (4096, 1.0), (8192, 1.0)):
    out = []
    for on in (1, 0, 1, 0):
        m = bench.make_mppi(K, hor, 0)
        m.setOption(capi.MPPI_OPT_BATCH_GRAPH, on)
        m.enqueueRngBatch(bench.X0, 42, 0, 300, st); torch.cuda.synchronize()
        t0 = time.perf_counter(); m.enqueueRngBatch(bench.X0, 42, 300, 2000, st); torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 2000 * 1e6)
        m.close()
    print(f"K={K} T={int(hor*100)}: graph {out[0]:.2f} / {out[2]:.2f} us per tick, plain launches {out[1]:.2f} / {out[3]:.2f}", flush=True)
