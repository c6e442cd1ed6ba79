"""How long the HOST takes to enqueue a block of plain-launch ticks (tbnav_mppi_enqueue_rng_batch returns when the launches are queued)
against the block's wall time with the wait: is the driver's 20-step block bound by the launch rate or by the device?
python tools/mppi_enqueue_host_cost.py [steps] [repeats]"""
import os, sys, time
node = os.environ.get("TBNAV_NODE")   # before anything of the runtime is loaded: its threads and first-touch pages inherit this
if node is not None:
    lo = 0 if node == "0" else 64
    os.sched_setaffinity(0, set(range(lo, lo + 64)) | set(range(lo + 128, lo + 192)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
bench.graft.load_package()
pin = os.environ.get("TBNAV_PIN")
if pin is not None:
    cpus = sorted(os.sched_getaffinity(0))
    os.sched_setaffinity(0, {cpus[int(pin) % len(cpus)]})
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 41
m = bench.make_mppi(1024, 0.5, 0)
torch.cuda.set_stream(torch.cuda.Stream(0))
st = torch.cuda.current_stream(0).cuda_stream
tk = 0
m.enqueueRngBatch(bench.X0, 42, tk, 50, st); tk += 50
torch.cuda.synchronize()
host, wall = [], []
for _ in range(reps):
    t0 = time.perf_counter()
    m.enqueueRngBatch(bench.X0, 42, tk, steps, st); tk += steps
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) / steps * 1e6); wall.append((t2 - t0) / steps * 1e6)
print(f"[node {node}] [pin {pin}, cpu now {os.sched_getcpu() if hasattr(os, 'sched_getcpu') else '?'}] {steps}-tick blocks x {reps}: host enqueue {np.median(host):.2f} us/tick (min {min(host):.2f}, max {max(host):.2f}); with the wait {np.median(wall):.2f} us/tick (min {min(wall):.2f}, max {max(wall):.2f})")
m.close()
