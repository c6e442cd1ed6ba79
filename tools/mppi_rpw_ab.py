"""Dev probe: rollouts per workgroup of the fused small-K kernel with the noise drawn inside (TBNAV_MPPI_OPT_KERNEL = -4 / -8 / -16):
K = 1024 ... 2048, ticks replayed from the batch call's graphs (2000 in a row, and blocks of 20 between synchronisations)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.load_package()
import bench
from rtn_amd import capi
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(dev))
st = torch.cuda.current_stream(dev).cuda_stream
for K, hor in ((1024, 0.5), (1536, 0.5), (2048, 0.5)):
    for r in (8, 4, 16, 8, 4):
        m = bench.make_mppi(K, hor, 0)
        m.setOption(capi.MPPI_OPT_KERNEL, -r)
        m.enqueueRngBatch(bench.X0, 42, 0, 300, st); torch.cuda.synchronize()
        t0 = time.perf_counter(); m.enqueueRngBatch(bench.X0, 42, 300, 2000, st); torch.cuda.synchronize()
        long_us = (time.perf_counter() - t0) / 2000 * 1e6
        tick, blocks = 2300, []
        for b in range(31):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.enqueueRngBatch(bench.X0, 42, tick, 20, st); torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / 20 * 1e6); tick += 20
        ms = m.kernelBreakdownMs(20) if hasattr(m, "kernelBreakdownMs") else None
        print(f"K={K} T={int(hor*100)} rollouts/workgroup {r:2d}: {long_us:.2f} us per tick over 2000, {sorted(blocks)[15]:.2f} median of 31 blocks of 20"
              + (f", kernels {[round(x * 1e3, 2) for x in ms]}" if ms else "") + f" ({m.rollout_kernel}, graph ticks {m.graphReplayedTicks()})", flush=True)
        m.close()
