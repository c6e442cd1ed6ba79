"""A-B of two builds of libtbnav_hip.so on the headline tick (K = 1024, T = 50, device noise): us per tick over 2000 ticks (graph replay)
and over 20-tick batches of plain launches (what the driver's --steps 20 times).  python tools/mppi_lib_ab.py <lib A> [lib B ...]
Each library is loaded in a process of its own."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.load_package()
    from rtn_amd import capi
    if sys.argv[2] != "default":
        capi.LIB_PATH = os.path.abspath(sys.argv[2])
    import torch, bench
    m = bench.make_mppi(1024, 0.5, 0)
    torch.cuda.set_stream(torch.cuda.Stream())
    st = torch.cuda.current_stream().cuda_stream
    tk = 0
    def run(n):
        global tk
        m.enqueueRngBatch(bench.X0, 42, tk, n, st); tk += n
    run(300); torch.cuda.synchronize()
    res = []
    for rep in range(5):
        t0 = time.perf_counter(); run(2000); torch.cuda.synchronize(); long_us = (time.perf_counter() - t0) / 2000 * 1e6
        short = []
        for q in range(30):
            run(5); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(20); torch.cuda.synchronize(); short.append((time.perf_counter() - t0) / 20 * 1e6)
        short.sort()
        res.append((long_us, short[len(short) // 2]))
    print(sys.argv[2], "us/tick over 2000 ticks:", " ".join(f"{a:.2f}" for a, _ in res), "| median of 20-tick batches:", " ".join(f"{b:.2f}" for _, b in res), flush=True)
    sys.exit(0)
for lib in sys.argv[1:] or ["default"]:
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], check=False)
