#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X hot paths (contract: see the round brief).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one MPPI control tick (controller::MPPI::newControls, mppi.cpp:72-140) over one batch
of synthetic control noise that is already resident in HBM when the timed region starts.
Workload at every N: BASELINE.json configs[1] per GPU — K=1024 rollouts, T=50 steps, shipped
controller parameters — so N>1 is weak scaling (global K = N*1024) with ONE all-gather of the
per-time-step soft-min records per tick (RCCL).  value = rollouts/s = N*K*steps / max-over-ranks
time.  Extra objects on the same JSON line:
  roofline        dominant kernel of the timed workload (mppi_rollout_cost) vs the HBM roofline
  roofline_large  same kernel set on BASELINE configs[3]'s per-call size on ONE GPU (K=65536, T=100),
                  where the path actually streams from HBM (315 MB algorithmic per tick)
  cpu_baseline    the oracle port (oracle/mppi_oracle.cpp, 1 core) on the same workload, rank 0, N=1
  rbpf            secondary headline: RBPF particle-updates/s (BASELINE configs[2]) when built
Only the cpu_baseline leg touches oracle/.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_ROLLOUT_STEP = 48.0          # rollout pass 16 B noise + 8 B J; weighting 8 B J + 16 B noise
BYTES_ROLLOUT_KERNEL = 24.0            # of which the rollout/cost kernel: reads duL,duR (16 B), writes J (8 B)

SHIPPED = dict(wheel_radius=0.033, wheel_base=0.16, lam=0.01, max_wheel_vel=6.35495, ul_var=0.9,
               ur_var=0.9, dt=0.01, Q=[1e4, 1e4, 1.0], R=[0.1, 0.1], P1=[1e3, 1e3, 1e3])
WAYPOINT = (1.0, 0.0, 1.5707)  # real_waypoints.yaml: waypoint 1
X0 = (0.0, 0.0, 0.0)


def make_mppi(K, horizon, device):
    from rtn_amd.mppi import MPPI, CartModel, LossFunc
    m = MPPI(CartModel(SHIPPED["wheel_radius"], SHIPPED["wheel_base"]),
             LossFunc(SHIPPED["Q"], SHIPPED["R"], SHIPPED["P1"]), SHIPPED["lam"], SHIPPED["max_wheel_vel"],
             SHIPPED["ul_var"], SHIPPED["ur_var"], horizon, SHIPPED["dt"], K, device)
    m.setWaypoint(*WAYPOINT)
    return m


def synth_noise(T, K, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    a = torch.randn(T, K, dtype=torch.float64, device=device, generator=g) * np.sqrt(SHIPPED["ul_var"])
    b = torch.randn(T, K, dtype=torch.float64, device=device, generator=g) * np.sqrt(SHIPPED["ur_var"])
    return a, b


def time_ticks(tick_fn, sync_fn, steps, warmup, barrier):
    for _ in range(warmup):
        tick_fn()
    sync_fn(); barrier(); sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        tick_fn()
    sync_fn(); barrier(); sync_fn()
    return time.perf_counter() - t0


def kernel_profile(m, a, b, stream, n):
    """Average per-kernel duration (ms), HIP events on the launch stream: each kernel of the tick launched back to back
    between one event pair (tbnav_mppi_profile_kernels) — a single launch of a 5 us kernel between two events measures
    the events as much as the kernel, and the rocprofv3 trace in profiles/ would not agree with it."""
    acc = np.zeros(3)
    n_tick = min(n, 50)
    for _ in range(n_tick):  # in tick order, one event pair per launch: right for kernels much longer than an event
        acc += np.array(m.profileTick(X0, a.data_ptr(), b.data_ptr(), stream))
    acc /= n_tick
    if acc[0] >= 0.02:
        return acc
    reps = max(2, (min(n, 200) // 2) * 2)
    acc = np.zeros(3)
    rounds = max(1, n // reps)
    for _ in range(rounds):
        acc += np.array(m.profileKernels(X0, a.data_ptr(), b.data_ptr(), stream, reps))
    return acc / rounds


def pmc_traffic(workload_key, kernel_prefix):
    """HBM bytes per launch of a kernel from the committed PMC passes (profiles/r01_traffic_pmc.json:
    separate FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 read correction calibrated on a known byte count)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic_pmc.json")) as f:
            wl = json.load(f)["workloads"][workload_key]
        for name, v in wl.items():
            if name.startswith(kernel_prefix):
                return v["hbm_bytes"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def roofline_obj(K, T, ms_kernels, ms_tick, kernel_name, traffic_key=None):
    alg = BYTES_ROLLOUT_KERNEL * K * T
    achieved = alg / (ms_kernels[0] * 1e-3) / 1e9
    tick_gbs = BYTES_PER_ROLLOUT_STEP * K * T / (ms_tick * 1e-3) / 1e9
    return {
        "bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
        "traffic": pmc_traffic(traffic_key, kernel_name.split("<")[0]) if traffic_key else None,
        "traffic_source": "profiles/r01_traffic_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" if traffic_key else None,
        "algorithmic_bytes_per_launch": alg,
        "kernel_ms": {"rollout": round(float(ms_kernels[0]), 6), "partials": round(float(ms_kernels[1]), 6),
                      "combine": round(float(ms_kernels[2]), 6)},
        "whole_tick": {"algorithmic_bytes": BYTES_PER_ROLLOUT_STEP * K * T, "ms": round(ms_tick, 6),
                       "achieved": round(tick_gbs, 3), "frac": round(tick_gbs / HBM_PEAK_GBS, 6)},
    }


def cpu_baseline(K, T, horizon, budget_s=12.0):
    """oracle port (kind 'port'), 1 core, same workload: ticks of K rollouts x T steps with state
    carried, noise pre-drawn (sampling excluded on both sides)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as orc
    d = dict(SHIPPED, horizon=horizon, rollouts=K)
    rng = np.random.default_rng(0)
    noise = rng.standard_normal((K, T, 2)) * np.sqrt(SHIPPED["ul_var"])
    u = np.zeros((2, T))
    orc.mppi_new_controls(d, u, (0, 0), WAYPOINT, X0, noise)  # warm
    n, t0 = 0, time.perf_counter()
    while True:
        r = orc.mppi_new_controls(d, u, (0, 0), WAYPOINT, X0, noise)
        u = r["u"]; n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 2000:
            break
    return {"value": round(n * K / el, 1), "unit": "rollouts/s", "cores": 1, "kind": "port",
            "sample": f"{n} ticks of K={K}, T={T} (oracle/mppi_oracle.cpp, g++ -O2, 1 thread, noise pre-drawn)",
            "ms_per_tick": round(el / n * 1e3, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large", action="store_true")
    ap.add_argument("--no-rbpf", action="store_true", help="skip the secondary RBPF object (development)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # dev switch: exercise the multi-rank path on a ONE-GPU box (all ranks on cuda:0, gloo exchange);
    # the real run is one rank per GPU over RCCL.
    one_gpu_test = os.environ.get("TBNAV_BENCH_ONE_GPU_GLOO") == "1"
    if one_gpu_test:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu_test:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    graft.load_package()
    from rtn_amd.sharded import HipShardBackend, ShardedMPPI

    K, horizon = 1024, 0.5  # BASELINE configs[1] per GPU
    m = make_mppi(K, horizon, local_rank)
    T = m.steps
    a, b = synth_noise(T, K, device, 1234 + rank)
    stream = torch.cuda.current_stream(device).cuda_stream

    if world == 1:
        def tick():
            m.enqueueDev(X0, a.data_ptr(), b.data_ptr(), stream)
        barrier = lambda: None  # noqa: E731
    else:
        sm = ShardedMPPI(HipShardBackend(m, device))

        def tick():
            sm.tick(X0, (a.data_ptr(), b.data_ptr()))

        def barrier():
            dist.barrier()

    sync = lambda: torch.cuda.synchronize(device)  # noqa: E731
    el = time_ticks(tick, sync, args.steps, args.warmup, barrier)
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cpu" if one_gpu_test else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    out_controls = m.lastControls(stream)
    assert all(np.isfinite(out_controls)), out_controls

    if rank == 0:
        ms_step = el / args.steps * 1e3
        value = world * K * args.steps / el
        ms_k = kernel_profile(m, a, b, stream, min(args.steps, 500))
        # one synchronous tick (launch + wait + 16-byte D2H), the latency a control loop sees
        t0 = time.perf_counter()
        for _ in range(200):
            m.newControlsDev(X0, a.data_ptr(), b.data_ptr(), stream)
        sync_ms = (time.perf_counter() - t0) / 200 * 1e3
        line = {
            "metric": "MPPI rollouts/s", "value": round(value, 1), "unit": "rollouts/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 6),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"MPPI newControls K={K} per GPU, T={T} (BASELINE configs[1]); global K={world * K}",
                       "noise": "resident in HBM, [T][K] fp64 x2", "state_carried": True,
                       "parallelism": f"rollout-shard x{world}" + (", 1 all-gather of soft-min records/tick" if world > 1 else "")},
            "rollout_steps_per_s": round(value * T, 1),
            "sync_tick_ms": round(sync_ms, 6),
            "roofline": roofline_obj(K, T, ms_k, ms_step, m.rollout_kernel, "mppi_K1024_T50" if (K, T) == (1024, 50) else None),
        }
        if world == 1 and not args.no_large:
            KL, HL = 65536, 1.0  # BASELINE configs[3] per-call size, on one GPU
            ml = make_mppi(KL, HL, local_rank)
            al, bl = synth_noise(ml.steps, KL, device, 99)
            tl = lambda: ml.enqueueDev(X0, al.data_ptr(), bl.data_ptr(), stream)  # noqa: E731
            el_l = time_ticks(tl, sync, 50, 10, lambda: None)
            ms_l = kernel_profile(ml, al, bl, stream, 50)
            rl = roofline_obj(KL, ml.steps, ms_l, el_l / 50 * 1e3, ml.rollout_kernel, "mppi_K65536_T100")
            rl["workload"] = f"MPPI newControls K={KL}, T={ml.steps} on 1 GPU"
            rl["rollouts_per_s"] = round(KL * 50 / el_l, 1)
            line["roofline_large"] = rl
            ml.close()
        if world == 1:
            # the exact-arc dynamics option, same workload, same timing as `value`
            ma = make_mppi(K, horizon, local_rank)
            ma.setDynamics("arc")
            el_a = time_ticks(lambda: ma.enqueueDev(X0, a.data_ptr(), b.data_ptr(), stream), sync, min(args.steps, 1000),
                              min(args.warmup, 100), lambda: None)
            n_a = min(args.steps, 1000)
            line["options"] = {"mppi_exact_arc_dynamics": {"rollouts_per_s": round(K * n_a / el_a, 1),
                                                           "ms_per_step": round(el_a / n_a * 1e3, 6)}}
            ma.close()
            # production tick: fresh perturbations drawn on the device every tick (Philox, inside the fused kernel) —
            # what a controller that does not bring its own noise pays; `value` is quoted with the inputs resident
            tk = [0]

            def prod_tick():
                m.enqueueRng(X0, 42, tk[0], stream)
                tk[0] += 1
            el_p = time_ticks(prod_tick, sync, n_a, min(args.warmup, 100), lambda: None)
            line["options"]["mppi_tick_with_device_noise"] = {"rollouts_per_s": round(K * n_a / el_p, 1),
                                                              "ms_per_step": round(el_p / n_a * 1e3, 6)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(K, T, horizon)
        if world == 1 and not args.no_rbpf:
            try:
                import bench_rbpf
                line["rbpf"] = bench_rbpf.run(device, args, with_cpu=not args.no_cpu_baseline)
            except ImportError:
                pass
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
