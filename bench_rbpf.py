"""RBPF leg of bench.py: BASELINE configs[2] — 1000 particles, 360-beam synthetic scan, 400x400 @ 0.05 m,
k = 50 samples round the mode (shipped slam.launch), ICP-ok branch, a 20-scan trajectory in a room small enough
that EVERY one of the 360 beams returns inside [range_min, range_max) at every pose.

Two modes, side by side (`modes`):
  reference_equal  TBNAV_RBPF_DF_REFERENCE — the reference's own brushfire field bit for bit (lazy, host cores; csrc/ref_field.hpp):
                   synchronous tbnav_rbpf_slam calls, device noise, first scan untimed;
  query_default    the C-ABI's default (exact nearest-obstacle distance per lookup): particle-updates/s = N * SLAM calls / wall time
                   of the logged run replayed through tbnav_rbpf_slam_batch with the standard normals drawn ON the device (nothing
                   but the 1.4 KB scan crosses PCIe).  Two of the timed scans are forced to RESAMPLE (skewed weights set beforehand,
                   untimed): their table copies, the state gather and the tile clones of the scan that follows are in the time.
roofline (dominant kernel = the raycast / log-odds update of the query mode): algorithmic bytes are COUNTED on the device
(TBNAV_RBPF_OPT_COUNT_CELLS): every distinct cell a scan writes is one f64 read + one f64 write = 16 B; the kernel's time by live HIP
events (`frac`) and by the committed rocprofv3 row of exactly this instantiation (`frac_rocprof`); `traffic` from the committed PMC passes.
Everything else (options, other rooms and shapes, noise forms, reference-field variants): bench_rbpf_detail.py, with bench.py --detail.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rbpf_cases import ROOM_BENCH, ROOM_SURVEY, TRAJ_BENCH as TRAJ_INC, TRAJ_SURVEY  # noqa: E402,F401  (pure constants; the rooms the full-size parity tests use)
RESAMPLE_AT = (8, 14)                 # timed scans forced to resample


def _world():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rbpf_cases as rc
    return rc


def _room_scan(pose, rng, walls, n_beams=360, sigma=0.01, beam_delta_deg=1.0):
    th, x, y = pose
    ang = th + np.deg2rad(beam_delta_deg) * np.arange(n_beams)
    c, s = np.cos(ang), np.sin(ang)
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = np.where(c > 0, (walls[1] - x) / c, np.where(c < 0, (walls[0] - x) / c, np.inf))
        ty = np.where(s > 0, (walls[3] - y) / s, np.where(s < 0, (walls[2] - y) / s, np.inf))
    return (np.minimum(tx, ty) + rng.normal(0.0, sigma, n_beams)).astype(np.float32)


def _skew(pf, N):
    w = np.full(N, 0.2 / N); w[N // 7] += 0.5; w[(5 * N) // 7] += 0.3
    pf.setParticles(w=w / w.sum())


def _kernel_threads(name):   # "rbpf_raycast_box<512, 6, false, 8>" -> 512
    return int(name.split("<")[1].rstrip(">").split(",")[0])


def reference_field_mode(device, N, k, map_half, walls, n_scans, host_threads=0, spread=None, start=(0.0, 0.0, 0.0), inc=None):
    """The product in the mode that reproduces the reference's distance field bit for bit (TBNAV_RBPF_DF_REFERENCE: what
    bmapping::ParticleFilter defaults to up to 4096 particles): the priority-queue brushfires run on the host's cores — ONE per
    distinct (particle state, the scan's insert / erase sequence), shared by the particles that are copies of one another and saw
    the same cells change (csrc/ref_field.hpp) — everything else on the device.  Synchronous calls, device noise, first scan
    untimed.  spread: a sampling spread (theta, x, y) instead of the shipped 1e-10 / 1e-8 / 1e-8 (slam.launch:24-26) — wide enough
    and every particle's beams end in other cells: nothing is shared, every particle pays for its own brushfire."""
    from rtn_amd import capi
    from rtn_amd.rbpf import ParticleFilter, default_params
    rc = _world()
    kw = {} if spread is None else {"sample_range": list(spread)}
    pf = ParticleFilter(default_params(N=N, k=k, map_min=-map_half, map_max=map_half, device=device.index or 0, pose0=start, **kw), df_mode="reference")
    pf.setOption(capi.RBPF_OPT_HOST_THREADS, host_threads)
    pf.setSeed(2026)
    steps, poses = rc.trajectory(n_scans, inc=inc or (TRAJ_INC if map_half > 5 else (0.03, 0.02, 0.01)), start=start)
    rng = np.random.default_rng(7)
    t, n = 0.0, 0
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        scan = _room_scan(poses[s], rng, walls)
        t0 = time.perf_counter()
        st = pf.SLAM(scan, u, cur, prev, True, t_icp, None)
        if s == 0:
            fires0 = pf.referenceFieldCounts()[2]
        if s >= 1:
            t += time.perf_counter() - t0; n += 1
    distinct, _, fires = pf.referenceFieldCounts()
    lazy = pf.referenceFieldStats()
    pf.close()
    return {"workload": f"RBPF N={N}, k={k}, {int(st.n_valid_beams)} valid beams of 360, {int(2 * map_half / 0.05)}^2 @0.05 m, distance field = the reference's brushfire (bit-exact mode)"
                        + ("" if spread is None else f", sampling spread {tuple(spread)} instead of the shipped 1e-10 / 1e-8 / 1e-8")
                        + f"; trajectory from {tuple(start)} by {tuple(inc or (TRAJ_INC if map_half > 5 else (0.03, 0.02, 0.01)))} per scan",
            "particle_updates_per_s": round(N * n / t, 1), "ms_per_scan": round(t / n * 1e3, 3), "scans_timed": n,
            "brushfires_per_scan": round((fires - fires0) / n, 1), "distinct_particle_states_at_the_end": distinct,
            "lazy_brushfire": lazy, "iterations_per_brushfire": round(lazy["iterations"] / max(lazy["passes"], 1), 1),
            "host_threads": host_threads or effective_cores()}


def workload(n_scans=20):
    rc = _world()
    steps, poses = rc.trajectory(n_scans, inc=TRAJ_INC)
    rng = np.random.default_rng(7)
    scans = [_room_scan(poses[s], rng, ROOM_BENCH) for s in range(n_scans)]
    return steps, scans


def run(device, args, N=1000, k=50, n_scans=20, with_cpu=True, detail=False):
    """What the driver's command runs: the headline replay, the counted-cells and event-timing passes its roofline needs, the
    reference-equal mode at configs[2], the two CPU baselines.  detail=True adds every other leg (bench_rbpf_detail.add_legs)."""
    import bench_profiles as bp
    from rtn_amd import capi
    from rtn_amd.rbpf import ParticleFilter, default_params
    mk = lambda: ParticleFilter(default_params(N=N, k=k, map_min=-10.0, map_max=10.0, device=device.index or 0))  # noqa: E731
    steps, scans = workload(n_scans)
    # ---- counted cells: their own pass (the counters cost device time)
    pf_c = mk()
    pf_c.setSeed(2026); pf_c.setOption(capi.RBPF_OPT_COUNT_CELLS, 1)
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        if s == 2:
            pf_c.scanCounts(reset=True)
        pf_c.SLAM(scans[s], u, cur, prev, True, t_icp, None)
    upd, distinct = pf_c.scanCounts()
    pf_c.close()
    n_counted = (n_scans - 2) * N
    upd_per, distinct_per = upd / n_counted, distinct / n_counted
    # ---- per-kernel durations: their own pass (each HIP event costs device time).  Three kinds of scan: plain; the one that
    #      resamples (its own gather launch); the FIRST scan after a resampling, whose map update makes every written tile of a shared
    #      map private — the roofline's algorithmic bytes are the plain scan's, so its kernel time is the plain scans' too
    pf_k = mk()
    pf_k.setSeed(2026); pf_k.setTiming(True)
    kms, kms_res, kms_cow, n_k, n_kr, n_kc = {}, {}, {}, 0, 0, 0
    after_resample = False
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        if s in RESAMPLE_AT:
            _skew(pf_k, N)
        st = pf_k.SLAM(scans[s], u, cur, prev, True, t_icp, None)
        k_propose, k_raycast, _ = pf_k.lastKernelNames()
        if s >= 4:  # (the map update's LDS array has adapted to the boxes' need by then: the steady state's kernels)
            tgt = kms_res if st.resampled else (kms_cow if after_resample else kms)
            for key, v in pf_k.kernelMs().items():
                tgt[key] = tgt.get(key, 0.0) + v
            n_kr += int(bool(st.resampled)); n_kc += int(after_resample and not st.resampled); n_k += int(not st.resampled and not after_resample)
        after_resample = bool(st.resampled)
    pf_k.close()
    kms = {key: v / max(n_k, 1) for key, v in kms.items() if key not in ("edt", "occupancy")}   # (query mode has no distance-field pass)
    kms_res = {key: v / max(n_kr, 1) for key, v in kms_res.items() if key not in ("edt", "occupancy")}
    kms_cow = {key: v / max(n_kc, 1) for key, v in kms_cow.items()}
    # ---- headline pass: the logged run replayed through tbnav_rbpf_slam_batch (the same synchronous per-scan calls, made from C),
    #      in stretches between the points where the weights are skewed (untimed) to force a resample
    pf = mk()
    pf.setSeed(2026)
    odom = np.array([steps[0][0]] + [st_[1] for st_ in steps], dtype=np.float64)
    u_all = np.array([st_[3] for st_ in steps], dtype=np.float64)
    ticp_all = np.array([st_[2] for st_ in steps], dtype=np.float64)
    t_total, n_timed, resamples, scan_ms = 0.0, 0, 0, []
    cuts = sorted(set([0, 2, n_scans] + [r for r in RESAMPLE_AT if r < n_scans]))
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        if lo in RESAMPLE_AT:
            _skew(pf, N)       # untimed: the filter's weights are made skewed, the timed stretch's first call does the resampling
        t0 = time.perf_counter()
        sts = pf.SLAMBatch(np.stack(scans[lo:hi]), u_all[lo:hi], odom[lo:hi + 1], ticp_all[lo:hi])
        dt = time.perf_counter() - t0
        if lo >= 2:  # first two scans: empty maps / first-touch tile allocation
            t_total += dt; n_timed += hi - lo; scan_ms.append(round(dt / (hi - lo) * 1e3, 4))
        resamples += sum(x.resampled for x in sts)
    ms_scan = t_total / n_timed * 1e3
    cap, free, tile_bytes = pf.poolStats()
    Bv, xsize, ysize, G = int(sts[-1].n_valid_beams), pf.xsize, pf.ysize, pf.G
    pf.close()
    # ---- the mode that reproduces the reference's field
    rm = reference_field_mode(device, N, k, 10.0, ROOM_BENCH, n_scans)
    # ---- the roofline of the dominant kernel: the committed profiler rows of exactly this instantiation and grid (N particles'
    #      workgroups + the normalise workgroup)
    dev_ms = sum(kms.values())
    alg_dom = distinct_per * 16.0 * N             # bytes the raycast launch has to move: one RMW per distinct cell
    t_rc = kms["raycast"] * 1e-3                  # s per launch, live HIP events
    wl = "rbpf_N1000_k50_400x400_plain_scans_only"
    rows = lambda name: None if "<" not in name or N != 1000 else (   # noqa: E731
        bp.rocprof_row(name, _kernel_threads(name) * (N + 1), wl) or bp.rocprof_row(name, _kernel_threads(name) * (N + 1)) or bp.rocprof_row(name, _kernel_threads(name) * N))
    rp_row, rp_propose = rows(k_raycast), rows(k_propose)
    pmc = bp.pmc_row(wl, k_raycast) if N == 1000 else None
    frac = lambda us: round(alg_dom / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 6)   # noqa: E731
    out = {
        "metric": "RBPF particle-updates/s", "value": round(N / (ms_scan * 1e-3), 1), "unit": "particle-updates/s",
        "value_is_for_mode": "query_default",
        # the two modes side by side, with equal weight (round-3 review): the one that reproduces the reference's results, and the
        # one `value` is measured in
        "modes": {
            "reference_equal": {"distance_field": "the reference's brushfire, bit for bit, run lazily on the host's cores (csrc/ref_field.hpp)",
                                "particle_updates_per_s": rm["particle_updates_per_s"], "ms_per_scan": rm["ms_per_scan"],
                                "parity": "<= 1e-9 / integers identical to the oracle, nothing injected (tests/test_rbpf_field_gpu.py): meets north_star's 1e-5",
                                "vs_target_1e5": round(rm["particle_updates_per_s"] / 1e5, 4)},
            "query_default": {"distance_field": "exact squared distance to the nearest occupied cell, per lookup, from the occupancy bits",
                              "particle_updates_per_s": round(N / (ms_scan * 1e-3), 1), "ms_per_scan": round(ms_scan, 4),
                              "parity": "<= 1e-9 to the restated filter over the EXACT distance; 7.5e-3 from the reference's field at 400x400: outside 1e-5",
                              "vs_target_1e5": round(N / (ms_scan * 1e-3) / 1e5, 2)}},
        "config": {"workload": f"RBPF SLAM N={N}, k={k}, {Bv} valid beams of 360, {xsize}x{ysize} @0.05 m, ICP-ok branch (BASELINE configs[2])",
                   "scans_timed": n_timed, "resamples": resamples, "entry_point": "tbnav_rbpf_slam_batch (synchronous per scan)",
                   "inputs": "standard normals drawn on the device (Philox); only the 1.4 KB scan is a host buffer", "room": list(ROOM_BENCH)},
        "ms_per_scan": round(ms_scan, 4), "scan_ms_by_stretch": scan_ms,
        "device_ms_per_scan": round(dev_ms, 4), "device_only_updates_per_s": round(N / (dev_ms * 1e-3), 1),
        "kernel_ms": {key: round(v, 4) for key, v in kms.items()},
        "kernel_ms_resampling_scan": {key: round(v, 4) for key, v in kms_res.items()},
        "kernel_ms_first_scan_after_a_resampling": {key: round(v, 4) for key, v in kms_cow.items()},
        "kernel_ms_scans": {"plain": n_k, "resampling": n_kr, "after_a_resampling": n_kc},
        "tile_pool": {"tiles": cap, "in_use": cap - free, "tile_bytes": tile_bytes,
                      "log_odds_bytes_in_use": (cap - free) * tile_bytes, "dense_equivalent_bytes": N * G * 8},
        "dtype": "f64+u16",
        "distance_field_mode": "query",
        "reference_field_mode": {"configs2_1000_particles_400x400": rm},
        # Dominant kernel = the map update.  What has to MOVE is one read-modify-write per DISTINCT cell of a scan (the kernel merges
        # the touches of one scan in LDS): `achieved` / `frac` are those bytes over the kernel's live HIP-event time; `traffic` is what
        # the PMC passes measured moving per launch (plain scans).
        "roofline": {"bound": "hbm", "kernel": k_raycast,
                     "achieved": round(alg_dom / t_rc / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": frac(kms["raycast"] * 1e3), "frac_events": frac(kms["raycast"] * 1e3),
                     # (ONE convention for every frac_rocprof: the named row's AVERAGE — the row of this workload's own plain-scan pass)
                     "frac_rocprof": None if rp_row is None else frac(rp_row["avg_us"]), "frac_rocprof_of": "avg_us",
                     "frac_rocprof_median": None if rp_row is None or not rp_row.get("median_us") else frac(rp_row["median_us"]),
                     "rocprof": rp_row,
                     "algorithmic_bytes_per_launch": round(alg_dom, 1), "distinct_cells_per_particle_and_scan": round(distinct_per, 1),
                     "touches_per_particle_and_scan": round(upd_per, 1),
                     "traffic": None if pmc is None else pmc["hbm_bytes"],
                     "traffic_source": None if pmc is None else pmc["source"] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; plain scans, no tile clones)",
                     "traffic_over_algorithmic": None if pmc is None else round(pmc["hbm_bytes"] / alg_dom, 3),
                     "traffic_reads_writes": None if pmc is None else [pmc.get("read_bytes"), pmc.get("write_bytes")],
                     "kernel_ms": round(kms["raycast"], 6),
                     "second_kernel": {"kernel": k_propose, "kernel_ms": round(kms["propose"], 6), "rocprof": rp_propose}},
    }
    if detail:
        import bench_rbpf_detail
        bench_rbpf_detail.add_legs(out, device, args, N, k, n_scans, dict(steps=steps, scans=scans, odom=odom, u_all=u_all, ticp_all=ticp_all, upd_per=upd_per,
                                                                            distinct_per=distinct_per, kms=kms, dev_ms=dev_ms, t_rc=t_rc, pmc=pmc, xsize=xsize))
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(k, scans, steps, threads=1, n_scans=11 if detail else 7)
        out["cpu_baseline_all_cores"] = cpu_baseline(k, scans, steps, threads=effective_cores())
    return out


def effective_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container on a
    256-thread host is often limited to a few cores; spawning 256 OpenMP threads there measures the scheduler)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(float(parts[0]) / float(parts[1]) + 0.5)))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, int(q / int(g.read().split()[0]) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(k, scans, steps, threads=1, n_scans=11):
    """oracle port (restated GridMapper + ParticleFilter incl. the reference's priority-queue brushfire; bit-exact vs
    the reference's GridMapper), same world / parameters, a BOUNDED sample (~5-10 s of CPU work): 2 particles per thread (min 32) x 10 timed
    scans.  threads > 1: the particle loop under OpenMP (the generous baseline)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as orc
    n_particles = max(32, 2 * threads)
    orc.lib().orc_set_threads(int(threads))
    try:
        pf = orc.PfAPI(orc.pf_params(N=n_particles, k=k, map_min=-10.0, map_max=10.0))
        nn = pf.normals_per_scan(True)
        t_total, n = 0.0, 0
        for s in range(n_scans):
            prev, cur, t_icp, u = steps[s]
            nz = np.random.default_rng(100 + s).standard_normal(nn)
            t0 = time.perf_counter()
            pf.slam(scans[s], u, cur, prev, True, t_icp, nz, trace=False)
            dt = time.perf_counter() - t0
            if s >= 1:
                t_total += dt; n += 1
        pf.close()
    finally:
        orc.lib().orc_set_threads(1)
    return {"value": round(n_particles * n / t_total, 2), "unit": "particle-updates/s", "cores": int(threads), "kind": "port",
            "cpu": _cpu_model(),
            "sample": f"{n_particles} particles x {n} scans, k={k}, 360 beams, 400x400 (oracle/rbpf_oracle.cpp, eager brushfire, g++ -O2, {threads} thread(s))",
            "ms_per_particle_update": round(t_total / (n_particles * n) * 1e3, 3)}
