#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X hot paths (contract: see the round brief).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one MPPI control tick (controller::MPPI::newControls, mppi.cpp:72-140) INCLUDING its Gaussian control-noise
sampling (mppi.cpp:173-184), which is part of the path: the perturbations of every tick are drawn on the device inside
the rollout kernel.  Only the 2*T warm-start controls are resident when the timed region starts.
Workload at every N: BASELINE.json configs[1] per GPU — K=1024 rollouts, T=50 steps, shipped controller parameters — so
N>1 is weak scaling (global K = N*1024) with ONE exchange of the per-time-step soft-min records per tick, issued by
libtbnav_hip.so itself (tbnav_mppi_attach_comm: shard partials -> exchange -> combine on the tick's stream; the ticks of the
timed region are enqueued by one C call per rank, no Python and no host synchronisation between them).  Between the GPUs of one
node the exchange is DIRECT when every rank can (tbnav_mppi_exchange_kind 2: each rank stores its records into every peer's
buffer over xGMI, the combine polls its own); otherwise an ncclAllGather.  `exchange` says which ran, and
`weak_tick_via_comm_all_gather` times the same tick through the all-gather beside it.
value = rollouts/s = N*K*steps / max-over-ranks time.  Extra objects on the same JSON line:
  roofline        dominant kernel of the timed workload vs the HBM roofline (per-kernel times: HIP events on the launch stream)
  latency_floor   what two dependent launches cost by themselves (the K=1024 tick is latency-bound, not HBM-bound)
  roofline_large  the kernel set of BASELINE configs[3]'s per-call size on ONE GPU (K=65536, T=100, resident noise),
                  where the path actually streams from HBM (315 MB algorithmic per tick)
  options         the same tick with resident noise; the exact-arc dynamics option
  cpu_baseline / cpu_baseline_all_cores   the oracle port (oracle/mppi_oracle.cpp) on 1 core / all host cores (OpenMP)
  rbpf            secondary headline: RBPF particle-updates/s (BASELINE configs[2]) per distance-field mode, bench_rbpf.py (+ bench_rbpf_detail.py with --detail)
  N > 1 only:     weak_tick_via_comm_all_gather, strong_scaling_configs3 (K=65536 split N ways), rbpf_sharded (1000 particles
                  per rank with cross-rank particle migration in the timed region) — under a watchdog: if one of them does not
                  come back, the line is printed without them (`multi_gpu_legs` says why)
Only the cpu_baseline legs touch oracle/.
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
             SHIPPED["ul_var"], SHIPPED["ur_var"], horizon, SHIPPED["dt"], K, device, keep_j=False)
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


def kernel_profile(m, a, b, stream, n, rng=None):
    """Average per-kernel duration (ms), HIP events on the launch stream: each kernel of the tick launched back to back
    between one event pair (tbnav_mppi_profile_kernels[_rng]) — a single launch of a 5 us kernel between two events measures
    the events as much as the kernel.  rng = (seed, tick): the PRODUCTION tick's kernels, i.e. the fused kernel's in-kernel-noise
    instantiation where that is what the timed region ran (round-3 review: the profile launched the resident-noise one)."""
    if rng is None:
        acc = np.zeros(3)
        n_tick = min(n, 50)
        for _ in range(n_tick):  # in tick order, one event pair per launch: right for kernels much longer than an event
            acc += np.array(m.profileTick(X0, a.data_ptr(), b.data_ptr(), stream))
        acc /= n_tick
        if acc[0] >= 0.02:
            return acc
    # (20 launches per event pair however many steps the run has: a queue of hundreds of launches runs each of them slower —
    #  4.3 us at 20 deep, 5.9 at 200 — and the figure should not depend on --steps)
    reps = max(2, (min(n, 20) // 2) * 2)
    acc = np.zeros(3)
    rounds = max(1, min(n // reps, 25))
    for r in range(rounds):
        if rng is None:
            acc += np.array(m.profileKernels(X0, a.data_ptr(), b.data_ptr(), stream, reps))
        else:
            acc += np.array(m.profileKernelsRng(X0, rng[0], rng[1] + r * reps, stream, reps))
    return acc / rounds


def roofline_obj(K, T, ms_kernels, ms_tick, kernel_name, grid_threads=None, traffic_key=None, stats_workload=None, sq_key=None):
    """The dominant kernel against the HBM roofline.  `achieved` / `frac` (= frac_events): SURVEY 8-d's algorithmic bytes of the
    kernel / its live HIP-event duration in THIS run; `frac_rocprof`: the same bytes / the rocprofv3 average of the row named in
    `rocprof` (profiles/, committed) — every figure recomputable from one named row.  The two clocks differ for the 5 us kernels
    (events: launch-to-launch interval of back-to-back launches; rocprofv3: begin-to-end of each dispatch while the tracer
    spaces the launches out — cold caches, idle clocks), within a few per cent for the long ones."""
    import bench_profiles as bp
    alg = BYTES_ROLLOUT_KERNEL * K * T
    achieved = alg / (ms_kernels[0] * 1e-3) / 1e9
    tick_gbs = BYTES_PER_ROLLOUT_STEP * K * T / (ms_tick * 1e-3) / 1e9
    row = bp.rocprof_row(kernel_name, grid_threads, stats_workload)
    pmc = bp.pmc_row(traffic_key, kernel_name) if traffic_key else None
    sq = bp.sq_row(sq_key, kernel_name) if sq_key else None
    frac_rp = None if row is None else round(alg / (row["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 6)
    return {
        "bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
        "frac_events": round(achieved / HBM_PEAK_GBS, 6), "frac_rocprof": frac_rp, "rocprof": row,
        "traffic": None if pmc is None else pmc["hbm_bytes"],
        "traffic_source": None if pmc is None else pmc["source"] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
        "algorithmic_bytes_per_launch": alg,
        "frac_rocprof_of": "avg_us",   # (every frac_rocprof of this line divides by the named row's AVERAGE; its median rides in `rocprof`)
        "sq_counters": sq,
        "kernel_ms": {"rollout": round(float(ms_kernels[0]), 6), "partials": round(float(ms_kernels[1]), 6),
                      "combine": round(float(ms_kernels[2]), 6)},
        "whole_tick": {"algorithmic_bytes": BYTES_PER_ROLLOUT_STEP * K * T, "ms": round(ms_tick, 6),
                       "achieved": round(tick_gbs, 3), "frac": round(tick_gbs / HBM_PEAK_GBS, 6)},
    }


def cpu_baseline(K, T, horizon, budget_s=12.0, threads=1):
    """oracle port (kind 'port'), same workload: ticks of K rollouts x T steps with state carried, noise pre-drawn
    (sampling excluded on the CPU side).  threads > 1: the rollout loop under OpenMP."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as orc
    orc.lib().orc_set_threads(int(threads))
    try:
        return _cpu_baseline(orc, K, T, horizon, budget_s, threads)
    finally:
        orc.lib().orc_set_threads(1)


def _cpu_baseline(orc, K, T, horizon, budget_s, threads):
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
        if el > budget_s or n >= 6000:
            break
    import bench_rbpf
    return {"value": round(n * K / el, 1), "unit": "rollouts/s", "cores": int(threads), "kind": "port", "cpu": bench_rbpf._cpu_model(),
            "sample": f"{n} ticks of K={K}, T={T} (oracle/mppi_oracle.cpp, g++ -O2, {threads} thread(s), noise pre-drawn)",
            "ms_per_tick": round(el / n * 1e3, 4)}


LINE_LIMIT = 6144   # bytes: the driver parses the LAST stdout line and keeps an 8 KB tail (round 5's 22 KB line came back unparsed)
DETAIL_PATH = os.environ.get("TBNAV_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")


def _pick(d, keys):
    return None if d is None else {k: d.get(k) for k in keys if k in d}


def _short(text, n=120):
    return text if text is None or len(text) <= n else text[:n - 1] + "\u2026"


def _cpu_obj(c):
    if c is None:
        return None
    o = _pick(c, ("value", "unit", "cores", "kind"))
    o["sample"] = _short(c.get("sample"), 140)
    return o


def _roof_obj(r):
    """The contract's roofline object, numbers and names only: bound, kernel, achieved / peak / frac (live HIP events), frac_rocprof
    with the row it divides by, traffic with its source, the algorithmic bytes."""
    if r is None:
        return None
    o = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_rocprof", "traffic", "algorithmic_bytes_per_launch", "traffic_over_algorithmic"))
    o["rocprof"] = _pick(r.get("rocprof"), ("source", "row", "avg_us"))
    o["traffic_source"] = None if not r.get("traffic_source") else r["traffic_source"].split(" (")[0]
    if isinstance(r.get("kernel_ms"), dict):
        o["kernel_us"] = {k: round(v * 1e3, 3) for k, v in r["kernel_ms"].items()}
    elif r.get("kernel_ms") is not None:
        o["kernel_us"] = round(r["kernel_ms"] * 1e3, 3)
    if r.get("traffic") and r.get("algorithmic_bytes_per_launch") and "traffic_over_algorithmic" not in o:
        o["traffic_over_algorithmic"] = round(r["traffic"] / r["algorithmic_bytes_per_launch"], 3)
    return o


def compact_line(full):
    """The record the driver parses: the contract's keys, one roofline and one cpu_baseline object, a small RBPF summary — numbers and
    names, no prose.  Everything else the run measured is in bench_detail.json (`detail`)."""
    keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_min", "ms_per_step_max", "repeats",
            "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: full.get(k) for k in keys}
    cfg = full.get("config") or {}
    out["config"] = {"workload": _short(cfg.get("workload")), "noise": _short(cfg.get("noise")), "parallelism": _short(cfg.get("parallelism"))}
    out["roofline"] = _roof_obj(full.get("roofline"))
    out["cpu_baseline"] = _cpu_obj(full.get("cpu_baseline"))
    out["cpu_baseline_all_cores"] = _cpu_obj(full.get("cpu_baseline_all_cores"))
    for k in ("sync_tick_ms", "graph_replayed_ticks", "exchange", "multi_gpu_legs"):
        if full.get(k) is not None:
            out[k] = _short(full[k]) if isinstance(full[k], str) else full[k]
    rb = full.get("rbpf")
    if rb is not None:
        modes = rb.get("modes") or {}
        out["rbpf"] = {
            "metric": rb.get("metric"), "unit": rb.get("unit"), "value": rb.get("value"), "value_is_for_mode": rb.get("value_is_for_mode"),
            "ms_per_scan": rb.get("ms_per_scan"), "dtype": rb.get("dtype"),
            "config": {"workload": _short((rb.get("config") or {}).get("workload"), 160)},
            "modes": {name: _pick(mo, ("particle_updates_per_s", "ms_per_scan", "vs_target_1e5", "max_rel_err_vs_reference")) for name, mo in modes.items()},
            "roofline": _roof_obj(rb.get("roofline")),
            "cpu_baseline": _cpu_obj(rb.get("cpu_baseline")), "cpu_baseline_all_cores": _cpu_obj(rb.get("cpu_baseline_all_cores")),
        }
    for k in ("strong_scaling_configs3", "rbpf_sharded", "weak_tick_via_comm_all_gather"):   # N > 1 legs: the figures only
        if full.get(k) is not None:
            out[k] = _pick(full[k], ("rollouts_per_s", "particle_updates_per_s", "ms_per_step", "ms_per_scan", "exchange_kind", "exchange", "resamples", "scaling"))
    out["detail"] = os.path.basename(DETAIL_PATH)
    return out


def emit(full):
    """bench_detail.json <- everything measured; stdout's LAST line <- the compact record (asserted under LINE_LIMIT bytes)."""
    try:
        with open(DETAIL_PATH, "w") as f:
            json.dump(full, f, indent=1)
            f.write("\n")
    except OSError as e:   # (a read-only checkout: the line still goes out)
        print(f"[bench] could not write {DETAIL_PATH}: {e}", file=sys.stderr, flush=True)
    text = json.dumps(compact_line(full), separators=(",", ":"))
    if len(text) >= LINE_LIMIT:   # never print a line the driver cannot read: drop the secondary object first
        c = compact_line(full)
        c.pop("rbpf", None)
        text = json.dumps(c, separators=(",", ":"))
    sys.stdout.flush()
    print(text, flush=True)


LEGS_BUDGET_S = 240.0  # the N > 1 legs take seconds (60 ticks, 14 scans, RCCL's first point-to-point connections)


class LegWatchdog:
    """Bounds the optional multi-GPU legs: if they are not through after `seconds`, rank 0 prints the headline line it already
    has (with the reason) and every rank leaves with os._exit — a hung collective cannot be interrupted any other way."""

    def __init__(self, rank, line, seconds):
        import threading
        self.rank, self.line = rank, line
        self.timer = threading.Timer(seconds, self._fire, args=(f"not finished after {seconds:.0f} s",))
        self.timer.daemon = True
        self.timer.start()

    def _fire(self, why):
        if self.rank == 0 and self.line is not None:
            out = dict(self.line)
            out["multi_gpu_legs"] = f"skipped: {why}"
            emit(out)
        else:
            time.sleep(3.0)  # (rank 0's line first)
        os._exit(0)

    def expire_now(self, why):
        self.timer.cancel()
        self._fire(why)

    def cancel(self):
        self.timer.cancel()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large", action="store_true")
    ap.add_argument("--no-rbpf", action="store_true", help="skip the secondary RBPF object (development)")
    ap.add_argument("--detail", action="store_true", help="every optional leg as well (per-shape roofline objects, options, noise forms, "
                    "reference-field variants, configs[4] as written); all of it goes to bench_detail.json, never into the last line")
    ap.add_argument("--repeats", type=int, default=31, help="the --steps block is timed this many times; ms_per_step is the median")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # dev switch of THIS SCRIPT (not of the library): exercise the multi-rank path on a ONE-GPU box (all ranks on
    # cuda:0, gloo exchange); the real run is one rank per GPU over RCCL.
    one_gpu_test = os.environ.get("TBNAV_BENCH_ONE_GPU_GLOO") == "1"
    if one_gpu_test:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        # a launcher that gives every rank ONE visible device (HIP_VISIBLE_DEVICES per process): that device is ordinal 0
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_gpu_test:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    graft.load_package()
    from rtn_amd.sharded import HipShardBackend, ShardedMPPI
    comm = None
    # dev switch: the round-2 exchange driven from Python (rtn_amd.sharded) instead of the library's own
    py_exchange = os.environ.get("TBNAV_BENCH_PY_EXCHANGE") == "1"
    transport = "ipc" if one_gpu_test else "rccl"   # ranks sharing one device: RCCL refuses them, the library's IPC transport does not
    if world > 1 and not py_exchange:
        # the communicator the LIBRARY exchanges through (include/tbnav_comm.h): rank 0's id travels over the job's own
        # process group; torch.distributed is left with the barriers and the max-over-ranks of the timing
        from rtn_amd.comm import Comm
        red_dev = "cpu" if one_gpu_test else device
        try:
            comm = Comm.from_torch_distributed(local_rank, transport=transport)
            comm.selftest(1 << 16)   # one all-gather + one ring of sends through the library's transport, checked, before anything is timed
            ok = torch.ones(1, device=red_dev)
        except Exception as e:  # noqa: BLE001 — the line says which exchange ran ("exchange"); the Python path is the round-2 one
            print(f"[bench rank {rank}] in-library communicator unavailable ({e}); using the torch.distributed exchange", file=sys.stderr, flush=True)
            comm, ok = None, torch.zeros(1, device=red_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)   # all ranks take the same path
        if float(ok.item()) == 0.0:
            comm = None

    K, horizon = 1024, 0.5  # BASELINE configs[1] per GPU
    m = make_mppi(K, horizon, local_rank)
    m.setRngShard(rank * K, world * K)
    T = m.steps
    a, b = synth_noise(T, K, device, 1234 + rank)
    # a stream of its own (not the legacy default stream): tbnav_mppi_enqueue_rng_batch replays captured hipGraphs of ticks, and a
    # capture cannot be taken on the null stream; every launch, event and synchronisation below is on / of this stream
    torch.cuda.set_stream(torch.cuda.Stream(device))
    stream = torch.cuda.current_stream(device).cuda_stream
    SEED = 42
    tk = [0]

    # THE TIMED STEP = the production tick: the Gaussian control-noise sampling is part of the path (north_star), so
    # the perturbations of every tick are drawn on the device, inside the fused rollout kernel (Philox); nothing is
    # resident beforehand but the 2*T warm-start controls.
    if world == 1:
        def tick():
            m.enqueueRng(X0, SEED, tk[0], stream)
            tk[0] += 1

        def ticks(n):  # the same n ticks enqueued by ONE call through the C boundary (tbnav_mppi_enqueue_rng_batch): from
            m.enqueueRngBatch(X0, SEED, tk[0], n, stream)  # Python an enqueue costs 8-10 us, as much as the tick itself
            tk[0] += n
        barrier = lambda: None  # noqa: E731
    elif comm is not None:
        m.attachComm(comm)   # every tick entry point of the handle is now the sharded tick, inside the library

        def tick():
            m.enqueueRng(X0, SEED, tk[0], stream)
            tk[0] += 1

        def ticks(n):
            m.enqueueRngBatch(X0, SEED, tk[0], n, stream)
            tk[0] += n

        def barrier():
            dist.barrier()
    else:  # dev switch: all ranks on one GPU over gloo (RCCL refuses two ranks on one device): the Python exchange path
        sm = ShardedMPPI(HipShardBackend(m, device))

        def tick():
            sm.tick(X0, ("rng", SEED, tk[0]))
            tk[0] += 1

        def barrier():
            dist.barrier()

    sync = lambda: torch.cuda.synchronize(device)  # noqa: E731
    graph_ticks_timed = 0
    els = None
    if world == 1:
        ticks(args.warmup)
        sync()  # (one rank: the barrier of the bracket is empty)
        g0 = m.graphReplayedTicks()
        # the block of EXACTLY --steps ticks, timed --repeats times back to back (state carried): 20 ticks are 0.2 ms of work, one
        # sample of that is inside the noise of a wake-up — the line reports the MEDIAN block (min / max beside it)
        els = []
        for _ in range(max(1, args.repeats)):
            t0 = time.perf_counter()
            ticks(args.steps)
            sync()
            els.append(time.perf_counter() - t0)
        el = float(np.median(els))
        graph_ticks_timed = (m.graphReplayedTicks() - g0) // max(1, args.repeats)
        el_py = time_ticks(tick, sync, args.steps, args.warmup, barrier) if args.detail else None  # one Python call per tick, for comparison
    elif comm is not None:
        els_box = []

        def headline():
            ticks(args.warmup)
            blocks = []
            for _ in range(max(1, args.repeats)):   # every block bracketed as the contract says; max over ranks per block, median of those
                sync(); barrier(); sync()
                t0 = time.perf_counter()
                ticks(args.steps)
                sync(); barrier(); sync()
                blocks.append(time.perf_counter() - t0)
            tb = torch.tensor(blocks, dtype=torch.float64, device="cpu" if one_gpu_test else device)
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)
            blocks = [float(x) for x in tb.cpu()]
            el = float(np.median(blocks))
            els_box[:] = blocks
            try:
                good = bool(np.all(np.isfinite(m.lastControls(stream))))
            except Exception as e:  # noqa: BLE001 — a direct exchange whose bound expired reports here
                print(f"[bench rank {rank}] headline ticks failed: {e}", file=sys.stderr, flush=True)
                good = False
            flag = torch.tensor([1.0 if good else 0.0], device="cpu" if one_gpu_test else device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)   # every rank knows whether every rank's ticks came through
            return el, float(flag.item()) == 1.0
        el, good = headline()
        if not good and m.exchangeKind() == 2:
            # the direct exchange passed its self-test and failed in the run: the same ticks through the communicator's all-gather
            print(f"[bench rank {rank}] direct exchange failed in the run; repeating the headline through the communicator's all-gather", file=sys.stderr, flush=True)
            m.attachComm(None); m.setDirectExchange(0); m.setInitialControls(0.0, 0.0); m.attachComm(comm)
            el, good = headline()
        assert good, "the sharded ticks did not come through"
        el_py = None
        els = list(els_box)
    else:
        el = time_ticks(tick, sync, args.steps, args.warmup, barrier)
        el_py = None
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cpu" if one_gpu_test else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    out_controls = m.lastControls(stream)
    assert all(np.isfinite(out_controls)), out_controls

    line = None
    if rank == 0:
        ms_step = el / args.steps * 1e3
        value = world * K * args.steps / el
        # (local launches of this rank's kernels — no exchange — also on a handle with a communicator attached)
        # the production tick's own kernels: the in-kernel-noise instantiation the timed region ran
        ms_k = kernel_profile(m, a, b, stream, min(args.steps, 500), rng=(SEED, 20_000_000))
        k_rollout, k_combine = m.lastKernelNames()
        sampler = "fp64 Box-Muller on 52-bit uniforms" if k_rollout.endswith(", 2>") else "fp32 Box-Muller on 24-bit uniforms"
        line = {
            "metric": "MPPI rollouts/s", "value": round(value, 1), "unit": "rollouts/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 6),
            "ms_per_step_min": None if not els else round(min(els) / args.steps * 1e3, 6),
            "ms_per_step_max": None if not els else round(max(els) / args.steps * 1e3, 6),
            "repeats": None if not els else len(els), "ms_per_step_is": "median over `repeats` timed blocks of `steps` ticks",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"MPPI newControls K={K} per GPU, T={T} (BASELINE configs[1]); global K={world * K}",
                       "noise": f"drawn in the rollout kernel, inside the timed tick: Philox4x32-10 + {sampler}",
                       "state_carried": True,
                       "parallelism": f"rollout-shard x{world}" + (", 1 exchange of soft-min records/tick" if world > 1 else "")},
            "rollout_steps_per_s": round(value * T, 1),
            "sync_tick_ms": None,   # (measured below, under the watchdog when N > 1)
            "entry_point": ("tbnav_mppi_enqueue_rng_batch" if (world == 1 or comm is not None) else "tbnav_mppi_shard_* per tick from Python (rtn_amd.sharded)"),
            # what actually ran in a timed block: whole chunks of 100 ticks are replayed from a captured hipGraph, the rest (all of
            # them when --steps < 100, as with the driver's --steps 20) are plain launches
            "graph_replayed_ticks": graph_ticks_timed,
            "exchange": None if world == 1 else (("direct peer stores (tbnav_mppi_exchange_kind 2)" if m.exchangeKind() == 2 else
                                                  ("ncclAllGather" if transport == "rccl" else "all-gather (IPC transport: ranks share one device)"))
                                                 + ", issued by libtbnav_hip.so on the tick's stream (tbnav_mppi_attach_comm)"
                                                 if comm is not None else f"torch.distributed {dist.get_backend()} all-gather from Python (rtn_amd.sharded)"),
            "ms_per_step_one_python_call_per_tick": None if el_py is None else round(el_py / args.steps * 1e3, 6),
            "roofline": dict(roofline_obj(K, T, ms_k, ms_step, k_rollout, None, "mppi_K1024_T50" if (K, T) == (1024, 50) else None, "mppi_K1024_T50", "mppi_K1024_T50_device_noise"),
                             combine_kernel=k_combine),
            # the tick is two dependent launches: what the guide prices for that alone (MI355X_MICROARCH.md, "boundary" row)
            "latency_floor": {"dependent_launches_per_tick": 2, "boundary_us_each": [1.45, 1.9]},
        }

    # ---- N > 1: everything after the headline runs under a watchdog: the headline above is complete, and a collective that does
    #      not come back (an exchange that hangs on this node) must not take it along — on expiry rank 0 prints the line without
    #      what is missing (`multi_gpu_legs` says so) and every rank leaves.
    wd = LegWatchdog(rank, line, LEGS_BUDGET_S) if world > 1 else None
    # one synchronous tick (launch + wait + 16-byte D2H), the latency a control loop sees.  With a communicator attached the
    # tick contains the all-gather, so EVERY rank takes part (on rank 0 alone it would wait for peers that never call)
    n_sync = 200
    if world > 1:
        m.setInitialControls(0.0, 0.0)   # (rank 0's local kernel profile advanced ITS warm start: every rank starts from the same controls again)
        sync(); barrier(); sync()
    t0 = time.perf_counter()
    for i in range(n_sync):
        if world == 1 or comm is not None:
            m.newControlsRng(X0, SEED, 10_000_000 + i, stream)
        else:
            tick(); sync()
    sync_ms = (time.perf_counter() - t0) / n_sync * 1e3

    if rank == 0:
        line["sync_tick_ms"] = round(sync_ms, 6)
    if world > 1:
        extra = {}
        try:
            extra = multi_gpu_legs(world, rank, local_rank, device, one_gpu_test, stream, sync, barrier, comm)
        except Exception as e:  # noqa: BLE001 — the peers may be inside a collective this rank never reaches: leave through the watchdog
            print(f"[bench rank {rank}] multi-GPU legs failed: {e!r}", file=sys.stderr, flush=True)
            wd.expire_now(f"failed on rank {rank}: {e!r}")
        wd.cancel()
        if rank == 0:
            line.update(extra)

    if rank == 0:
        if world == 1 and args.detail and not args.no_large:
            KL, HL = 65536, 1.0  # BASELINE configs[3] per-call size, on one GPU
            ml = make_mppi(KL, HL, local_rank)
            al, bl = synth_noise(ml.steps, KL, device, 99)
            tl = lambda: ml.enqueueDev(X0, al.data_ptr(), bl.data_ptr(), stream)  # noqa: E731
            el_l = time_ticks(tl, sync, 50, 10, lambda: None)
            ms_l = kernel_profile(ml, al, bl, stream, 50)
            rl = roofline_obj(KL, ml.steps, ms_l, el_l / 50 * 1e3, ml.lastKernelNames()[0], KL, "mppi_K65536_T100", "mppi_K65536_T100", "mppi_K65536_T100")
            rl["workload"] = f"MPPI newControls K={KL}, T={ml.steps} on 1 GPU, noise resident in HBM (the streaming regime)"
            rl["rollouts_per_s"] = round(KL * 50 / el_l, 1)
            line["roofline_large"] = rl
            ml.close()
            # one GPU's share of BASELINE configs[3] as written (K = 65536 over 8 GPUs = 8192 rollouts, T = 100): the per-rank
            # compute of the strong-scaling leg, measured here (the exchange adds one all-gather of 100 x 4 records per tick)
            ms8 = make_mppi(KL // 8, HL, local_rank)
            a8, b8 = synth_noise(ms8.steps, KL // 8, device, 98)
            el8 = time_ticks(lambda: ms8.enqueueDev(X0, a8.data_ptr(), b8.data_ptr(), stream), sync, 200, 20, lambda: None)
            tk8 = [0]

            def rng8():
                ms8.enqueueRng(X0, SEED, tk8[0], stream)
                tk8[0] += 1
            el8r = time_ticks(rng8, sync, 200, 20, lambda: None)
            # its own roofline object (round-4 review): the production tick's kernels at this shape, HIP events; the committed rows are
            # those of this workload's own profiler passes (tools/profile_round.sh: mppi_K8192_T100)
            ms_k8 = kernel_profile(ms8, a8, b8, stream, 200, rng=(SEED, 30_000_000))
            rl8 = roofline_obj(KL // 8, ms8.steps, ms_k8, el8r / 200 * 1e3, ms8.lastKernelNames()[0], None, "mppi_K8192_T100", "mppi_K8192_T100", "mppi_K8192_T100")
            rl8["combine_kernel"] = ms8.lastKernelNames()[1]
            rl8["fp64_flops_note"] = "the rollout is ~200 fp64 operations per rollout-step: 8192 x 100 x 200 / kernel time against the 78.6 TFLOP/s vector peak is the other roof (DESIGN.md section 4)"
            line["configs3_shard_one_gpu"] = {"workload": f"MPPI K={KL // 8} (= 65536 / 8), T={ms8.steps}, one GPU", "kernel": ms8.rollout_kernel,
                                              "ms_per_step_resident_noise": round(el8 / 200 * 1e3, 6), "ms_per_step_device_noise": round(el8r / 200 * 1e3, 6),
                                              "rollouts_per_s_device_noise": round(KL // 8 * 200 / el8r, 1), "roofline": rl8}
            ms8.close()
        if world == 1 and args.detail:
            n_a = min(args.steps, 1000)
            # the same tick with the perturbations already resident in HBM ([T][K] fp64 x2): the parity-mode data flow
            el_r = time_ticks(lambda: m.enqueueDev(X0, a.data_ptr(), b.data_ptr(), stream), sync, n_a, min(args.warmup, 100), lambda: None)
            line["options"] = {"mppi_tick_with_resident_noise": {"rollouts_per_s": round(K * n_a / el_r, 1),
                                                                 "ms_per_step": round(el_r / n_a * 1e3, 6)}}
            # the exact-arc dynamics option, production tick
            ma = make_mppi(K, horizon, local_rank)
            ma.setDynamics("arc")
            tka = [0]

            def arc_tick():
                ma.enqueueRng(X0, SEED, tka[0], stream)
                tka[0] += 1
            el_a = time_ticks(arc_tick, sync, n_a, min(args.warmup, 100), lambda: None)
            line["options"]["mppi_exact_arc_dynamics"] = {"rollouts_per_s": round(K * n_a / el_a, 1),
                                                          "ms_per_step": round(el_a / n_a * 1e3, 6)}
            ma.close()
            # the same production tick with the fp32 sampler (TBNAV_MPPI_OPT_SAMPLER = 0, the default up to round 5): what the narrower sampler buys
            from rtn_amd import capi as _capi
            mw = make_mppi(K, horizon, local_rank)
            mw.setOption(_capi.MPPI_OPT_SAMPLER, 0)
            tkw = [0]

            def narrow_ticks(n):
                mw.enqueueRngBatch(X0, SEED, tkw[0], n, stream)
                tkw[0] += n
            narrow_ticks(min(args.warmup, 100)); sync()
            t0w = time.perf_counter()
            narrow_ticks(n_a); sync()
            el_w = time.perf_counter() - t0w
            line["options"]["mppi_tick_fp32_sampler"] = {"rollouts_per_s": round(K * n_a / el_w, 1), "ms_per_step": round(el_w / n_a * 1e3, 6),
                                                          "kernel": mw.lastKernelNames()[0]}
            mw.close()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(K, T, horizon, threads=1)
            import bench_rbpf
            line["cpu_baseline_all_cores"] = cpu_baseline(K, T, horizon, threads=bench_rbpf.effective_cores(), budget_s=8.0)
        if world == 1 and not args.no_rbpf:
            import bench_rbpf
            line["rbpf"] = bench_rbpf.run(device, args, with_cpu=not args.no_cpu_baseline, detail=args.detail)
        emit(line)
    if world > 1:
        # (the line is out: a teardown that does not come back — a peer already gone, a communicator that waits — must not keep the
        #  job alive: after a minute every rank leaves)
        import threading
        bye = threading.Timer(60.0, lambda: os._exit(0))
        bye.daemon = True
        bye.start()
        # handles before the communicator they exchange through, the communicator before the job's own process group
        m.close()
        if comm is not None:
            comm.close()
        dist.barrier()
        dist.destroy_process_group()
        bye.cancel()


def multi_gpu_legs(world, rank, local_rank, device, one_gpu_test, stream, sync, barrier, comm=None):
    """Two more measurements when N > 1 (same processes, after the headline):
      strong_scaling_configs3  BASELINE configs[3]: K = 65536, T = 100 split N ways (K/N rollouts per rank, resident
                               noise — the streaming regime), one all-gather of records per tick;
      rbpf_sharded             BASELINE configs[2] per rank (1000 particles each, weak): slam_local + ONE all-gather of
                               the weights + the global normalise/selection on every rank; two of the timed scans are
                               forced to resample, so the cross-rank particle migration (tile blobs over RCCL) is timed."""
    from rtn_amd.sharded import HipRbpfShardBackend, HipShardBackend, ShardedMPPI, ShardedRBPF
    out = {}
    red_dev = "cpu" if one_gpu_test else device
    if comm is not None:
        # ---- the headline's tick once more with the records carried by the communicator's all-gather (RCCL) instead of the direct
        #      stores into the peers' buffers: what the exchange costs either way, same processes, same K per rank
        mb = make_mppi(1024, 0.5, local_rank)
        mb.setDirectExchange(False)
        mb.attachComm(comm)
        mb.enqueueRngBatch(X0, 42, 0, 10, stream)
        sync(); barrier(); sync()
        t0 = time.perf_counter()
        mb.enqueueRngBatch(X0, 42, 10, 100, stream)
        sync(); barrier(); sync()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # the two exchanges by themselves (no rollouts): 200 rounds of the 3.2 KB-per-rank record block each way
        probe = {"all_gather_us": round(mb.exchangeProbe(200, stream), 3), "all_gather_kind": mb.exchangeKind()}
        out["weak_tick_via_comm_all_gather"] = {"workload": "the headline's tick (K=1024 per rank, T=50, device noise), 100 ticks, exchange kind 1",
                                                "exchange_kind": mb.exchangeKind(), "ms_per_step": round(float(t.item()) / 100 * 1e3, 6),
                                                "rollouts_per_s": round(world * 1024 * 100 / float(t.item()), 1)}
        mb.close()
        mp = make_mppi(1024, 0.5, local_rank)
        mp.attachComm(comm)
        if mp.exchangeKind() == 2:
            probe["direct_us"] = round(mp.exchangeProbe(200, stream), 3)
        mp.close()
        out["exchange_probe"] = dict(probe, note="microseconds per exchange of the K=1024 tick's record block (T*8 doubles per rank) alone, HIP events on rank 0's stream; "
                                                 "direct = publish + collect kernels (the tick's combine polls instead of the collect launch)")
    # ---- MPPI strong scaling
    KL, HL = 65536, 1.0
    ml = make_mppi(KL // world, HL, local_rank)
    al, bl = synth_noise(ml.steps, KL // world, device, 99 + rank)
    if comm is not None:
        ml.attachComm(comm)
        el = time_ticks(lambda: ml.enqueueDev(X0, al.data_ptr(), bl.data_ptr(), stream), sync, 50, 10, barrier)
    else:
        sml = ShardedMPPI(HipShardBackend(ml, device))
        el = time_ticks(lambda: sml.tick(X0, (al.data_ptr(), bl.data_ptr())), sync, 50, 10, barrier)
    t = torch.tensor([el], dtype=torch.float64, device="cpu" if one_gpu_test else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out["strong_scaling_configs3"] = {"workload": f"MPPI K={KL} total, T={ml.steps}, K/N = {KL // world} per rank, resident noise",
                                      "rollouts_per_s": round(KL * 50 / float(t.item()), 1), "ms_per_step": round(float(t.item()) / 50 * 1e3, 6),
                                      "exchange_kind": ml.exchangeKind() if comm is not None else None,
                                      "scaling": "strong"}
    ml.close()
    # ---- RBPF weak scaling with migration
    import bench_rbpf
    from rtn_amd.rbpf import ParticleFilter, default_params
    n_local, k = 1000, 50
    steps, scans = bench_rbpf.workload(14)
    pf = ParticleFilter(default_params(N=n_local, k=k, map_min=-10.0, map_max=10.0, device=local_rank))
    pf.setSeed(2026)   # one seed: every rank draws its slice of the ensemble's stream (tbnav_rbpf_set_rng_shard)
    if comm is not None:
        pf.setParticles(w=np.full(n_local, 1.0 / (n_local * world)))
        pf.attachComm(comm)   # pf.SLAM is now this rank's part of the sharded scan, exchanged inside the library
        sr = None
    else:
        sr = ShardedRBPF(HipRbpfShardBackend(pf, device))
    t_total, n_timed, resamples, t_res, t_plain = 0.0, 0, 0, [], []
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        if s in (6, 10):  # skew the GLOBAL weights: heavy particles on the first and the last rank
            w = np.full(n_local, 0.2 / (n_local * world))
            if rank == 0:
                w[n_local // 7] += 0.5
            if rank == world - 1:
                w[(5 * n_local) // 7] += 0.3
            pf.setParticles(w=w)
        sync(); barrier(); sync()
        t0 = time.perf_counter()
        if sr is None:
            st = pf.SLAM(scans[s], u, cur, prev, True, t_icp, None)
        else:
            st, _, _ = sr.tick(scans[s], u, cur, prev, True, t_icp, None, 3 * k + 3)
        sync(); barrier(); sync()
        dt = time.perf_counter() - t0
        if s >= 2:
            t_total += dt; n_timed += 1
            (t_res if st.resampled else t_plain).append(dt)
        resamples += st.resampled
    t = torch.tensor([t_total], dtype=torch.float64, device="cpu" if one_gpu_test else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out["rbpf_sharded"] = {"workload": f"RBPF N={n_local} per rank ({n_local * world} global), k={k}, 360 beams, 400x400; weights all-gather + "
                                       "global selection per scan, particle migration when resampling fires",
                           "particle_updates_per_s": round(n_local * world * n_timed / float(t.item()), 1),
                           "ms_per_scan": round(float(t.item()) / n_timed * 1e3, 4), "scans_timed": n_timed, "resamples": resamples,
                           "ms_per_scan_rank0": {"without_resample": round(float(np.mean(t_plain)) * 1e3, 4) if t_plain else None,
                                                 "resampling_with_migration": round(float(np.mean(t_res)) * 1e3, 4) if t_res else None},
                           "bytes_migrated_rank0": None if sr is None else sr.bytes_migrated, "scaling": "weak",
                           "exchange": (f"libtbnav_hip.so (tbnav_rbpf_attach_comm), transport {'RCCL' if comm.uses_rccl else 'IPC (ranks share one device)'}"
                                        if sr is None else f"torch.distributed {dist.get_backend()} (rtn_amd.sharded)")}
    pf.close()
    return out


if __name__ == "__main__":
    main()
