"""Dev probe: tick time of MPPI across ensemble sizes and kernel choices (TBNAV_MPPI_OPT_KERNEL: default / 0 sequential /
n > 0 time-parallel with n steps per thread / -8, -16 fused), resident noise and device noise:
python tools/mppi_size_sweep.py [horizon] [K ...]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
import bench
from rtn_amd import capi
from rtn_amd.mppi import MPPI, CartModel, LossFunc
hor = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
Ks = [int(a) for a in sys.argv[2:]] or [1024, 2048, 4096, 8192, 16384, 32768, 65536]
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
S = bench.SHIPPED
for K in Ks:
    for kern in [None, 0, "scan", 5, 10, 12, 16, -8, -16] if not os.environ.get("KERNS") else [None if k == "None" else (k if k == "scan" else int(k)) for k in os.environ["KERNS"].split(",")]:
        try:
            m = MPPI(CartModel(S["wheel_radius"], S["wheel_base"]), LossFunc(S["Q"], S["R"], S["P1"]), S["lam"], S["max_wheel_vel"],
                     S["ul_var"], S["ur_var"], hor, S["dt"], K, 0, keep_j=False, kernel=kern)
        except Exception as e:  # a choice the handle refuses for this shape
            print(f"K={K:6d} kernel option {kern}: refused ({type(e).__name__})", flush=True)
            continue
        m.setWaypoint(*bench.WAYPOINT)
        a, b = bench.synth_noise(m.steps, K, dev, 1)
        out = []
        for mode in ("dev", "rng"):
            f = (lambda i: m.enqueueDev(bench.X0, a.data_ptr(), b.data_ptr(), st)) if mode == "dev" else (lambda i: m.enqueueRng(bench.X0, 42, i, st))
            for i in range(20): f(i)
            torch.cuda.synchronize()
            n = 200
            t0 = time.perf_counter()
            for i in range(n): f(100 + i)
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / n * 1e6)
        print(f"K={K:6d} T={m.steps:3d} option {str(kern):5s} {m.rollout_kernel[:40]:40s} resident {out[0]:7.1f} us  device-noise {out[1]:7.1f} us", flush=True)
        m.close()
