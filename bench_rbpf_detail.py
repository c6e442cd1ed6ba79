"""The optional legs of the RBPF bench (bench.py --detail; never part of the driver's command): the options and SURVEY 8-d's other runs,
the SURVEY room and the configs[4] shapes as first-class roofline legs, where the device noise is drawn, the long replay, the
reference-field mode's variants, the map layout's traffic floor.  add_legs() fills them into the object bench_rbpf.run() built."""
from __future__ import annotations

import time

import numpy as np

from bench_rbpf import (HBM_PEAK_GBS, RESAMPLE_AT, ROOM_BENCH, ROOM_SURVEY, TRAJ_INC, TRAJ_SURVEY, _kernel_threads, _room_scan, _skew, _world,
                        reference_field_mode, workload)


def configs4_shard(device, N=12500, k=50, n_scans=8):
    """One GPU's shard of BASELINE configs[4] (100 000 particles / 8 GPUs, 2000 x 2000 cells @ 0.05 m, 1080-beam scans): the
    per-rank work of that configuration measured on this GPU — replayed through tbnav_rbpf_slam_batch, device noise, the
    first two scans (first-touch tile allocation) untimed.  Across ranks the per-scan exchange adds one all-gather of
    100 000 weights and the global selection."""
    from rtn_amd.rbpf import ParticleFilter, default_params
    rc = _world()
    bd = 1.0 / 3.0
    pf = ParticleFilter(default_params(N=N, k=k, map_min=-50.0, map_max=50.0, beam_delta_deg=bd, device=device.index or 0), pool_bytes=16 << 30)
    pf.setSeed(5)
    steps, poses = rc.trajectory(n_scans, inc=(0.05, 0.04, 0.03))
    rng = np.random.default_rng(8)
    scans = np.stack([_room_scan(poses[s], rng, (-3.0, 3.0, -2.5, 2.5), n_beams=1080, beam_delta_deg=bd) for s in range(n_scans)])
    odom = np.array([steps[0][0]] + [st_[1] for st_ in steps], dtype=np.float64)
    u_all = np.array([st_[3] for st_ in steps], dtype=np.float64)
    ticp_all = np.array([st_[2] for st_ in steps], dtype=np.float64)
    pf.SLAMBatch(scans[:2], u_all[:2], odom[:3], ticp_all[:2])
    t0 = time.perf_counter()
    sts = pf.SLAMBatch(scans[2:], u_all[2:], odom[2:], ticp_all[2:])
    dt = (time.perf_counter() - t0) / (n_scans - 2)
    cap, free, tile_bytes = pf.poolStats()
    out = {"workload": f"RBPF N={N} (= 100 000 / 8), k={k}, {int(sts[-1].n_valid_beams)} valid beams of 1080, {pf.xsize}x{pf.ysize} @0.05 m, one GPU",
           "particle_updates_per_s": round(N / dt, 1), "ms_per_scan": round(dt * 1e3, 4), "scans_timed": n_scans - 2,
           "resamples": int(sum(x.resampled for x in sts)), "log_odds_bytes_in_use": (cap - free) * tile_bytes,
           "dense_equivalent_bytes": N * pf.G * 8}
    pf.close()
    return out


def map_update_leg(device, label, N, k, map_half, walls, inc, n_scans=12, n_beams=360, beam_delta_deg=1.0, pool_bytes=0,
                   traffic_key=None, stats_workload=None, sq_key=None):
    """A first-class leg for ONE workload of the scan update (round-4 review: every BASELINE shape carries its own roofline): the
    kernels that ran (names as the profiler spells them), their HIP-event times over the plain scans, the distinct cells counted on
    the device (TBNAV_RBPF_OPT_COUNT_CELLS) -> algorithmic bytes of the map update, its fraction of the HBM roofline by events and
    by the AVERAGE of this workload's own committed profiler row, and the PMC traffic of this workload's own passes."""
    from rtn_amd import capi
    from rtn_amd.rbpf import ParticleFilter, default_params
    import bench_profiles as bp
    rc = _world()
    mk = lambda: ParticleFilter(default_params(N=N, k=k, map_min=-map_half, map_max=map_half, beam_delta_deg=beam_delta_deg,  # noqa: E731
                                               device=device.index or 0), pool_bytes=pool_bytes)
    steps, poses = rc.trajectory(n_scans, inc=inc)
    rng = np.random.default_rng(7)
    scans = [_room_scan(poses[s], rng, walls, n_beams=n_beams, beam_delta_deg=beam_delta_deg) for s in range(n_scans)]
    pf_c = mk()
    pf_c.setSeed(2026); pf_c.setOption(capi.RBPF_OPT_COUNT_CELLS, 1)
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        if s == 2:
            pf_c.scanCounts(reset=True)
        st = pf_c.SLAM(scans[s], u, cur, prev, True, t_icp, None)
    upd, distinct = pf_c.scanCounts()
    pf_c.close()
    distinct_per, upd_per = distinct / ((n_scans - 2) * N), upd / ((n_scans - 2) * N)
    pf_k = mk()
    pf_k.setSeed(2026); pf_k.setTiming(True)
    kms, n_k, wall = {}, 0, 0.0
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        t0 = time.perf_counter()
        st = pf_k.SLAM(scans[s], u, cur, prev, True, t_icp, None)
        if s >= 4:   # (the LDS array has adapted to the boxes' need by then)
            wall += time.perf_counter() - t0
            for key, v in pf_k.kernelMs().items():
                kms[key] = kms.get(key, 0.0) + v
            n_k += 1
    k_propose, k_raycast, _ = pf_k.lastKernelNames()
    cap, free, tile_bytes = pf_k.poolStats()
    pf_k.close()
    kms = {key: v / n_k for key, v in kms.items() if key not in ("edt", "occupancy")}
    alg = distinct_per * 16.0 * N
    t_rc = kms["raycast"] * 1e-3
    grid = _kernel_threads(k_raycast) * (N + 1) if "<" in k_raycast else None
    row = None if grid is None else (bp.rocprof_row(k_raycast, grid, stats_workload) or bp.rocprof_row(k_raycast, _kernel_threads(k_raycast) * N, stats_workload))
    pmc = bp.pmc_row(traffic_key, k_raycast) if traffic_key else None
    return {"workload": f"RBPF {label}: N={N}, k={k}, {int(st.n_valid_beams)} valid beams of {n_beams}, {int(2 * map_half / 0.05)}^2 @0.05 m, walls {list(walls)}, trajectory step {list(inc)}; "
                        f"synchronous scans with event timing, device noise",
            "kernels": {"propose": k_propose, "raycast": k_raycast},
            "kernel_ms": {key: round(v, 4) for key, v in kms.items()}, "scans_timed": n_k,
            "log_odds_bytes_in_use": (cap - free) * tile_bytes,
            "roofline": {"bound": "hbm", "kernel": k_raycast, "kernel_ms": round(kms["raycast"], 6),
                         "algorithmic_bytes_per_launch": round(alg, 1),
                         "algorithmic_bytes_note": f"{distinct_per:.1f} distinct cells written per particle and scan (counted on the device) x 16 B x N; per-touch count {upd_per:.1f}",
                         "achieved": round(alg / t_rc / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / t_rc / 1e9 / HBM_PEAK_GBS, 6), "frac_events": round(alg / t_rc / 1e9 / HBM_PEAK_GBS, 6),
                         "frac_rocprof": None if row is None else round(alg / (row["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 6),
                         "frac_rocprof_of": "avg_us", "rocprof": row,
                         "traffic": None if pmc is None else pmc["hbm_bytes"],
                         "traffic_source": None if pmc is None else pmc["source"] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; plain scans)",
                         "traffic_over_algorithmic": None if pmc is None else round(pmc["hbm_bytes"] / alg, 3),
                         "sq_counters": bp.sq_row(sq_key, k_raycast) if sq_key else None,
                         "second_kernel": {"kernel": k_propose, "kernel_ms": round(kms["propose"], 6),
                                           "rocprof": bp.rocprof_row(k_propose, _kernel_threads(k_propose) * (N + 1), stats_workload) if "<" in k_propose else None}}}


def noise_forms(device, N, k, n_scans=30):
    """Where the device noise is drawn (TBNAV_RBPF_OPT_NOISE_IN_KERNEL), side by side on the bench workload: 1 — inside
    rbpf_propose, the beam table through its leading workgroup, two launches per scan; 0 (default since round 6) — rbpf_sample_normals
    stores the stream first, three launches.  Wall time of synchronous calls without event timing, and the proposal kernel by HIP events."""
    from rtn_amd import capi
    from rtn_amd.rbpf import ParticleFilter, default_params
    steps, scans = workload(n_scans)
    out = {}
    for name, val in (("in_kernel", 1), ("stored_first", 0)):
        res = {}
        for timing in (False, True):
            pf = ParticleFilter(default_params(N=N, k=k, map_min=-10.0, map_max=10.0, device=device.index or 0))
            pf.setSeed(2026); pf.setTiming(timing); pf.setOption(capi.RBPF_OPT_NOISE_IN_KERNEL, val)
            wall, n, prop = 0.0, 0, 0.0
            for s, (prev, cur, t_icp, u) in enumerate(steps):
                t0 = time.perf_counter()
                pf.SLAM(scans[s], u, cur, prev, True, t_icp, None)
                if s >= 6:
                    wall += time.perf_counter() - t0; n += 1
                    if timing:
                        prop += pf.kernelMs()["propose"]
            if timing:
                res["propose_kernel_ms"] = round(prop / n, 5); res["propose_kernel"] = pf.lastKernelNames()[0]
            else:
                res["ms_per_synchronous_scan"] = round(wall / n * 1e3, 5)
            pf.close()
        out[name] = res
    return out


def layout_floor(walls, inc, n_scans=14, first=2, seed=7):
    """The HBM bytes per particle and scan the map update cannot go below WITH THIS MAP LAYOUT (no GPU needed): every 128-byte line that
    holds a touched cell is fetched whole, every 32-byte sector that holds one is written whole (profiles/r05_fetch_write_calibration.txt:
    what FETCH_SIZE / WRITE_SIZE count on gfx950 for partial-line 16-byte accesses), tiles of 32 x 32 cells whose rows are 256 contiguous
    bytes.  The cells of the scan's Bresenham rays from the robot's cell at the trajectory's poses (the particles are micrometres from
    them), as distinct cells, sectors and lines; mean over scans [first, n_scans)."""
    rc = _world()
    res, map_min, rmin, rmax = 0.05, -10.0, 0.12, 3.5
    _, poses = rc.trajectory(n_scans, inc=inc)
    rng = np.random.default_rng(seed)
    rows = []
    for s in range(n_scans):
        th, x, y = poses[s]
        scan = _room_scan(poses[s], rng, walls).astype(np.float64)
        ang = th + np.deg2rad(1.0) * np.arange(scan.size)
        ok = (scan >= rmin) & (scan < rmax)
        ex, ey = x + scan * np.cos(ang), y + scan * np.sin(ang)
        cx, cy = int(np.floor((x - map_min) / res)), int(np.floor((y - map_min) / res))
        cells = set()
        for b in np.flatnonzero(ok):
            x1, y1 = int(np.floor((ex[b] - map_min) / res)), int(np.floor((ey[b] - map_min) / res))
            x0, y0 = cx, cy
            dx, dy = abs(x1 - x0), abs(y1 - y0)
            sx, sy = (1 if x1 > x0 else -1), (1 if y1 > y0 else -1)
            err = dx - dy
            while True:   # (a textbook Bresenham: the counts move by a fraction of a percent between variants; the kernel's is grid_mapper.cpp:229-270's)
                cells.add((x0, y0))
                if x0 == x1 and y0 == y1:
                    break
                e2 = 2 * err
                if e2 > -dy:
                    err -= dy; x0 += sx
                if e2 < dx:
                    err += dx; y0 += sy
        if s >= first:
            rows.append((len(cells), len({(i, j >> 2) for i, j in cells}), len({(i, j >> 4) for i, j in cells}), int(ok.sum())))
    r = np.array(rows, dtype=np.float64).mean(axis=0)
    return {"distinct_cells": round(float(r[0]), 1), "sectors_32B": round(float(r[1]), 1), "lines_128B": round(float(r[2]), 1), "valid_beams": round(float(r[3]), 1),
            "algorithmic_bytes": float(r[0]) * 16.0, "read_floor_bytes": float(r[2]) * 128.0, "write_floor_bytes": float(r[1]) * 32.0}


def long_replay(device, N, k, n_scans=56, warm=8):
    """The bench workload replayed by ONE tbnav_rbpf_slam_batch call, as it comes (no weights skewed: the filter does not resample on it):
    what a scan costs when nothing sits between the launches — the headline cuts its replay into calls of 6 scans and forces a resampling
    at two of the cuts, which is where its ms_per_scan exceeds the kernels' sum."""
    from rtn_amd.rbpf import ParticleFilter, default_params
    steps, scans = workload(n_scans)
    odom = np.array([steps[0][0]] + [st_[1] for st_ in steps], dtype=np.float64)
    u_all = np.array([st_[3] for st_ in steps], dtype=np.float64)
    ticp_all = np.array([st_[2] for st_ in steps], dtype=np.float64)
    sc = np.stack(scans)
    pf = ParticleFilter(default_params(N=N, k=k, map_min=-10.0, map_max=10.0, device=device.index or 0))
    pf.setSeed(2026)
    pf.SLAMBatch(sc[:warm], u_all[:warm], odom[:warm + 1], ticp_all[:warm])
    t0 = time.perf_counter()
    sts = pf.SLAMBatch(sc[warm:], u_all[warm:], odom[warm:], ticp_all[warm:])
    dt = time.perf_counter() - t0
    k_propose, k_raycast, _ = pf.lastKernelNames()
    pf.close()
    n = n_scans - warm
    return {"ms_per_scan": round(dt / n * 1e3, 4), "particle_updates_per_s": round(N * n / dt, 1), "scans_timed": n, "calls": 1,
            "resamples": int(sum(x.resampled for x in sts)), "kernels": {"propose": k_propose, "raycast": k_raycast}}


def configs4_as_written(device, N=100_000, P=8, k=50, n_scans=7):
    """BASELINE configs[4] as written on ONE GPU: 100 000 particles in 8 shards of 12 500 (tbnav_rbpf_group, every member on this
    device: the library's own sharded scan — weights all-gather, global normalise beside the map update, migration when
    resampling fires — with its in-process copy transport; on 8 devices the same calls go through RCCL), 1080-beam scans,
    2000 x 2000 cells, device noise; one of the timed scans is forced to resample across members.  Beside it: ONE handle holding
    all 100 000 particles.  (8 members on one device run one after another: this prices the sharded code path, not a speed-up.)"""
    from rtn_amd.rbpf import ParticleFilter, ParticleFilterGroup, default_params
    rc = _world()
    bd = 1.0 / 3.0
    kw = dict(map_min=-50.0, map_max=50.0, beam_delta_deg=bd)
    steps, poses = rc.trajectory(n_scans, inc=(0.05, 0.04, 0.03))
    rng = np.random.default_rng(8)
    scans = [_room_scan(poses[s], rng, (-3.0, 3.0, -2.5, 2.5), n_beams=1080, beam_delta_deg=bd) for s in range(n_scans)]
    out = {}
    for name, mk in (("eight_shards", lambda: ParticleFilterGroup(default_params(N=N, k=k, **kw), [device.index or 0] * P, pool_bytes_per_member=10 << 30)),
                     ("one_handle", lambda: ParticleFilter(default_params(N=N, k=k, device=device.index or 0, **kw), pool_bytes=80 << 30))):
        pf = mk()
        pf.setSeed(5)
        plain, res = [], []
        for s, (prev, cur, t_icp, u) in enumerate(steps):
            if s == 4:
                w = np.full(N, 0.3 / N); w[7] += 0.3; w[60_000] += 0.3; w[N - 1] += 0.1
                pf.setParticles(w=w)
            t0 = time.perf_counter()
            st = pf.SLAM(scans[s], u, cur, prev, True, t_icp, None)
            dt = time.perf_counter() - t0
            if s >= 2:
                (res if st.resampled else plain).append(dt)
        out[name] = {"ms_per_scan_without_resample": round(float(np.mean(plain)) * 1e3, 3), "ms_per_resampling_scan": round(float(np.mean(res)) * 1e3, 3) if res else None,
                     "particle_updates_per_s": round(N / float(np.mean(plain)), 1)}
        pf.close()
    out["workload"] = f"RBPF N={N}, k={k}, 1080-beam scans, 2000x2000 @0.05 m (BASELINE configs[4]) on one GPU"
    return out


def add_legs(out, device, args, N, k, n_scans, ctx):
    from rtn_amd.rbpf import ParticleFilter, default_params
    steps, scans, odom, u_all, ticp_all = ctx["steps"], ctx["scans"], ctx["odom"], ctx["u_all"], ctx["ticp_all"]
    mk = lambda kk=k: ParticleFilter(default_params(N=N, k=kk, map_min=-10.0, map_max=10.0, device=device.index or 0))  # noqa: E731
    big = not getattr(args, "no_large", False)

    def sync_calls(pf_x, feed, first=2, last=None, skew=False):   # one tbnav_rbpf_slam call per scan from this harness
        t, n = 0.0, 0
        for s, (prev, cur, t_icp, u) in enumerate(steps[:last]):
            if skew and s in RESAMPLE_AT:
                _skew(pf_x, N)
            t0 = time.perf_counter()
            pf_x.SLAM(scans[s], u, cur, prev, True, t_icp, feed(s))
            if s >= first:
                t += time.perf_counter() - t0; n += 1
        pf_x.close()
        return {"value": round(N / (t / n), 1), "ms_per_scan": round(t / n * 1e3, 4)}

    def replay(pf_x, icp_ok):
        pf_x.setSeed(2026)
        icp = np.full(n_scans, 1 if icp_ok else 0, dtype=np.int32)
        pf_x.SLAMBatch(np.stack(scans[:2]), u_all[:2], odom[:3], ticp_all[:2], icp_ok=icp[:2])
        t0 = time.perf_counter()
        pf_x.SLAMBatch(np.stack(scans[2:]), u_all[2:], odom[2:], ticp_all[2:], icp_ok=icp[2:])
        dt = (time.perf_counter() - t0) / (n_scans - 2)
        pf_x.close()
        return {"value": round(N / dt, 1), "ms_per_scan": round(dt * 1e3, 4)}

    # parity mode: 1.2 MB/scan of reference-order normals copied H2D inside the call (PCIe-inclusive, never the headline)
    pf_h = mk()
    nn = pf_h.numNormals(True)
    out["host_normals"] = sync_calls(pf_h, lambda s: np.random.default_rng(100 + s).standard_normal(nn))
    pf_s = mk(); pf_s.setSeed(2026)
    out["single_calls_from_python"] = sync_calls(pf_s, lambda s: None, skew=True)
    pf_m = mk(); pf_m.setSeed(2026); pf_m.setScanMatching(True)
    out["options"] = {"scan_matching": sync_calls(pf_m, lambda s: None, last=12),          # per-particle hill climbing before sampling (not the reference)
                      "icp_failed_branch": replay(mk(), False),                               # every scan with icp_ok = 0 (particle_filter.cpp:157-176)
                      "k10": replay(mk(10), True)}                                            # num_samples_mode = 10 instead of the shipped 50
    out["noise_forms"] = noise_forms(device, N, k)
    out["long_replay_one_call_no_resampling"] = long_replay(device, N, k)
    rc_ = _world()
    rf = out["reference_field_mode"]
    rf["launch_configuration_40_particles_80x80"] = reference_field_mode(device, 40, 50, 2.0, rc_.ROOM_SMALL, 12)
    if big:
        # SURVEY 8-d's trajectory starts on a corner of four cells and moves by whole cells: the 1e-8 m sampling spread then DOES put beams
        # in different cells and little is shared.  Off the corners most particles see the same cells change: what the state sharing buys
        rf["configs2_off_the_cell_corners"] = reference_field_mode(device, N, k, 10.0, ROOM_BENCH, 8, start=(0.013, 0.0137, 0.0211), inc=(0.07, 0.0213, 0.0117))
        rf["configs2_every_particle_distinct"] = reference_field_mode(device, N, k, 10.0, ROOM_BENCH, 3, spread=(0.02, 0.05, 0.05))
        # SURVEY 8-d's own room (+-3.0 / +-2.5 m, 246 valid beams): its boxes do not fit four workgroups per CU, the map update runs another
        # instantiation — a first-class leg with its own kernel name, bytes, profiler row and PMC row (round-4 review)
        out["survey_room"] = map_update_leg(device, "SURVEY 8-d room", N, k, 10.0, ROOM_SURVEY, TRAJ_SURVEY, traffic_key="rbpf_N1000_k50_400x400_survey_room",
                                            stats_workload="rbpf_N1000_k50_400x400_survey_room", sq_key="rbpf_N1000_k50_400x400_survey_room")
        shard4 = configs4_shard(device)
        shard4["roofline_leg"] = map_update_leg(device, "configs[4] / 8", 12500, k, 50.0, ROOM_SURVEY, (0.05, 0.04, 0.03), n_scans=8, n_beams=1080,
                                                beam_delta_deg=1.0 / 3.0, pool_bytes=16 << 30, traffic_key="rbpf_N12500_2000x2000_1080beams",
                                                stats_workload="rbpf_N12500_2000x2000_1080beams", sq_key="rbpf_N12500_2000x2000_1080beams")
        out["configs4_shard_one_gpu"] = shard4
        out["configs4_as_written_one_gpu"] = configs4_as_written(device)
    # ---- more on the headline's roofline: what this map layout lets the traffic go down to, SURVEY's per-touch bytes, the whole update
    r, pmc, t_rc, kms, dev_ms = out["roofline"], ctx["pmc"], ctx["t_rc"], ctx["kms"], ctx["dev_ms"]
    fl = layout_floor(ROOM_BENCH, TRAJ_INC)
    fl_bytes = (fl["read_floor_bytes"] + fl["write_floor_bytes"]) * N
    r["layout_floor"] = {"bytes_per_launch": round(fl_bytes, 1), "over_algorithmic": round((fl["read_floor_bytes"] + fl["write_floor_bytes"]) / fl["algorithmic_bytes"], 3),
                         "reads": round(fl["read_floor_bytes"] * N, 1), "writes": round(fl["write_floor_bytes"] * N, 1),
                         "lines_128B": fl["lines_128B"], "sectors_32B": fl["sectors_32B"], "distinct_cells": fl["distinct_cells"],
                         "traffic_over_floor": None if pmc is None else round(pmc["hbm_bytes"] / fl_bytes, 3)}
    alg_ref = ctx["upd_per"] * 16.0 * N   # one RMW per (beam, cell) touch — bytes the kernel by design does NOT move (it merges touches in LDS)
    r["per_touch_bytes"] = {"bytes_per_launch": round(alg_ref, 1), "frac": round(alg_ref / t_rc / 1e9 / HBM_PEAK_GBS, 6)}
    # what one particle-update has to move on the device: the log-odds RMW, the slice of its occupancy bitmap the lookups read, its
    # normals, and the per-stage outputs the C-ABI keeps (trace)
    slice_bytes = min(ctx["xsize"], 2 * (int(np.ceil(3.5 / 0.05)) + 2 + 48) + 1) * 4 * 8
    dev_alg_per = ctx["distinct_per"] * 16.0 + slice_bytes + (3 * k + 3) * 8 + (k * 5 + 17) * 8
    r["whole_update"] = {"algorithmic_bytes_per_particle_update": round(dev_alg_per, 1), "achieved": round(dev_alg_per * N / (dev_ms * 1e-3) / 1e9, 3),
                         "frac": round(dev_alg_per * N / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)}
