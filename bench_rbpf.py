"""RBPF leg of bench.py: BASELINE configs[2] — 1000 particles, 360-beam synthetic scan, 400x400 @ 0.05 m,
k = 50 samples round the mode (shipped slam.launch), ICP-ok branch, 20-scan trajectory of SURVEY.md 8-d.

particle-updates/s = N * SLAM calls / wall time of the synchronous tbnav_rbpf_slam calls with the
standard normals drawn ON the device (normals == NULL: nothing but the 1.4 KB scan crosses PCIe).
`host_normals` is the parity-mode figure, where the 1.2 MB/scan of reference-order normals is a host
buffer copied inside the call (PCIe-inclusive; never the headline); `device_ms_per_scan` is the sum of
the kernels' HIP-event durations alone.

roofline: priced on the kernel that dominates the distance-field mode that ran (see run()); algorithmic
bytes per particle-update are SURVEY.md 8-d's  k*Bv*8 + (C_free+Bv)*16 + G_reach*16.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0


def _world():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rbpf_cases as rc
    return rc


def _room_scan(pose, rng, walls, n_beams=360, sigma=0.01):
    th, x, y = pose
    ang = th + np.deg2rad(1.0) * np.arange(n_beams)
    c, s = np.cos(ang), np.sin(ang)
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = np.where(c > 0, (walls[1] - x) / c, np.where(c < 0, (walls[0] - x) / c, np.inf))
        ty = np.where(s > 0, (walls[3] - y) / s, np.where(s < 0, (walls[2] - y) / s, np.inf))
    return (np.minimum(tx, ty) + rng.normal(0.0, sigma, n_beams)).astype(np.float32)


def run(device, args, N=1000, k=50, n_scans=20, with_cpu=True):
    from rtn_amd.rbpf import ParticleFilter, default_params
    rc = _world()
    pf = ParticleFilter(default_params(N=N, k=k, map_min=-10.0, map_max=10.0, device=device.index or 0))
    steps, poses = rc.trajectory(n_scans, inc=(0.07, 0.10, 0.05))
    rng = np.random.default_rng(7)
    scans = [_room_scan(poses[s], rng, rc.ROOM_SURVEY) for s in range(n_scans)]
    nn = pf.numNormals(True)
    normals = [np.random.default_rng(100 + s).standard_normal(nn) for s in range(n_scans)]
    # parity-mode pass first (host normals, PCIe-inclusive), on its own filter
    pf_h = ParticleFilter(default_params(N=N, k=k, map_min=-10.0, map_max=10.0, device=device.index or 0))
    t_host, n_host = 0.0, 0
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        t0 = time.perf_counter()
        pf_h.SLAM(scans[s], u, cur, prev, True, t_icp, normals[s])
        if s >= 2:
            t_host += time.perf_counter() - t0; n_host += 1
    pf_h.close()
    # per-kernel durations: their own pass on their own filter, with the C-ABI's event timing switched on (the
    # events cost device time themselves, so the headline pass below runs without them)
    pf_k = ParticleFilter(default_params(N=N, k=k, map_min=-10.0, map_max=10.0, device=device.index or 0))
    pf_k.setSeed(2026)
    pf_k.setTiming(True)
    kms, n_k = {}, 0
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        pf_k.SLAM(scans[s], u, cur, prev, True, t_icp, None)
        if s >= 2:
            n_k += 1
            for key, v in pf_k.kernelMs().items():
                kms[key] = kms.get(key, 0.0) + v
    pf_k.close()
    kms = {key: v / n_k for key, v in kms.items()}
    t_total = 0.0
    n_timed = 0
    resamples = 0
    pf.setSeed(2026)
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        t0 = time.perf_counter()
        st = pf.SLAM(scans[s], u, cur, prev, True, t_icp, None)
        dt = time.perf_counter() - t0
        if s >= 2:  # first two scans: empty maps / first-touch
            t_total += dt; n_timed += 1
        resamples += st.resampled
    ms_scan = t_total / n_timed * 1e3
    # the per-particle scan-matching option on the same scans, its own filter
    pf_m = ParticleFilter(default_params(N=N, k=k, map_min=-10.0, map_max=10.0, device=device.index or 0))
    pf_m.setSeed(2026)
    pf_m.setScanMatching(True)
    t_sm, n_sm = 0.0, 0
    for s, (prev, cur, t_icp, u) in enumerate(steps[:12]):
        t0 = time.perf_counter()
        pf_m.SLAM(scans[s], u, cur, prev, True, t_icp, None)
        if s >= 2:
            t_sm += time.perf_counter() - t0; n_sm += 1
    pf_m.close()
    dev_ms = sum(kms.values())
    G = pf.G
    Bv = int(st.n_valid_beams)
    c_free = 30 * Bv  # ~30 free cells per ray in this room
    alg_per_update = k * Bv * 8 + (c_free + Bv) * 16 + G * 16
    # Distance-field modes (TBNAV_RBPF_DF): "query" (default) answers each lookup from the occupancy bitmap — no
    # transform in the SLAM path; "window" refreshes a window per particle before the update; "full" transforms
    # the whole map after every update (the reference's data flow).  The roofline object prices the kernel that
    # dominates the mode that ran; `whole_update` keeps SURVEY.md 8-d's reference-data-flow bytes for the sum.
    mode = os.environ.get("TBNAV_RBPF_DF", "query")
    if os.environ.get("TBNAV_RBPF_FULL_EDT") == "1":
        mode = "full"
    if mode not in ("full", "window"):
        mode = "query"
    move = max(float(np.hypot(steps[-1][2][1], steps[-1][2][2])), abs(float(steps[-1][3][1])))
    half_cells = int(np.ceil((3.5 + move + 8.0 * np.sqrt(1e-8)) / 0.05)) + 3
    win_cells = min(2 * half_cells + 1, pf.xsize) ** 2
    traffic = None
    if mode == "query":
        # dominant kernel: the raycast / log-odds update.  Algorithmic bytes: every cell a beam touches is one
        # f64 read + one f64 write (SURVEY.md 8-d's (C_free + Bv) * 16 counts a cell once per touching beam; the
        # kernel merges the touches of one scan, so the bytes that have to move are the DISTINCT cells).
        dom_name = "rbpf_raycast_tile (log-odds update)"
        dom_ms = kms["raycast"]
        alg_dom = (c_free + Bv) * 16
        alg_note = "(C_free + Bv) * 16 B per particle, SURVEY.md 8-d"
        try:
            import json
            with open(os.path.join(ROOT, "profiles", "r01_traffic_pmc.json")) as f:
                wl = json.load(f)["workloads"]["rbpf_N1000_k50_400x400"]
            if N == 1000:
                traffic = wl["rbpf_raycast_tile"]["hbm_bytes"]
        except (OSError, KeyError, ValueError):
            pass
    else:
        dom_name = "rbpf_edt_compact (distance field" + ("" if mode == "full" else f", {int(np.sqrt(win_cells))}^2-cell window per particle") + ")"
        dom_ms = kms["occupancy"] + kms["edt"]
        alg_dom = (G if mode == "full" else win_cells) * 16
        alg_note = "16 B per refreshed cell"
        try:
            import json
            with open(os.path.join(ROOT, "profiles", "r01_traffic_pmc.json")) as f:
                wl = json.load(f)["workloads"]["rbpf_N1000_k50_400x400"]
            if N == 1000 and mode == "window":
                traffic = sum(v["hbm_bytes"] for name, v in wl.items() if name.startswith("rbpf_edt"))
        except (OSError, KeyError, ValueError):
            pass
    out = {
        "metric": "RBPF particle-updates/s", "value": round(N / (ms_scan * 1e-3), 1), "unit": "particle-updates/s",
        "config": {"workload": f"RBPF SLAM N={N}, k={k}, {Bv} valid beams, {pf.xsize}x{pf.ysize} @0.05 m, ICP-ok branch "
                               "(BASELINE configs[2])", "scans_timed": n_timed, "resamples": resamples,
                   "inputs": "standard normals drawn on the device (Philox); only the 1.4 KB scan is a host buffer"},
        "host_normals": {"value": round(N / (t_host / n_host), 1), "ms_per_scan": round(t_host / n_host * 1e3, 4),
                         "note": "parity mode: 1.2 MB/scan of reference-order normals copied H2D inside the call (PCIe-inclusive)"},
        "ms_per_scan": round(ms_scan, 4), "device_ms_per_scan": round(dev_ms, 4),
        "device_only_updates_per_s": round(N / (dev_ms * 1e-3), 1),
        "kernel_ms": {key: round(v, 4) for key, v in kms.items()},
        "dtype": "f64+u16",
        "options": {"scan_matching": {"value": round(N / (t_sm / n_sm), 1), "ms_per_scan": round(t_sm / n_sm * 1e3, 4),
                                      "note": "per-particle hill climbing on the likelihood field before sampling (not the reference)"}},
        "distance_field_mode": mode,
        "roofline": {"bound": "hbm", "kernel": dom_name,
                     "achieved": round(alg_dom * N / (dom_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg_dom * N / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": traffic,
                     "traffic_source": "profiles/r01_traffic_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
                     "algorithmic_bytes_per_launch": alg_dom * N, "algorithmic_bytes_note": alg_note,
                     "whole_update": {"algorithmic_bytes_per_particle_update": alg_per_update,
                                      "achieved": round(alg_per_update * N / (dev_ms * 1e-3) / 1e9, 3),
                                      "frac": round(alg_per_update * N / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                                      "note": "reference data flow (k*Bv*8 lookups + (C_free+Bv)*16 + G*16 transform) over the "
                                              "device time of one scan; > 1 means traffic the reference needs is not moved at all"}},
    }
    pf.close()
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(k, scans, steps)
    return out


def cpu_baseline(k, scans, steps, n_particles=16, n_scans=6):
    """oracle port (restated GridMapper + ParticleFilter, bit-exact vs the reference's GridMapper), 1 core,
    same world / parameters, a BOUNDED sample: 16 particles x 6 scans."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as orc
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
    return {"value": round(n_particles * n / t_total, 2), "unit": "particle-updates/s", "cores": 1, "kind": "port",
            "sample": f"{n_particles} particles x {n} scans, k={k}, 360 beams, 400x400 (oracle/rbpf_oracle.cpp incl. the "
                      "reference's priority-queue brushfire, g++ -O2, 1 thread)",
            "ms_per_particle_update": round(t_total / (n_particles * n) * 1e3, 3)}
