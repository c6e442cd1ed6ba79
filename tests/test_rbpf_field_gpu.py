"""R9 (SURVEY.md section 8-a, section 7 hard part 1): the distance field WITHOUT injecting the oracle's.

Two contracts, both end to end through the C-ABI with nothing injected:
  (ii)  default mode (exact nearest-obstacle query): the effect of the exact-EDT-vs-brushfire difference on eta,
        normalised weights, Neff and the best pose is measured against the oracle (the reference's brushfire, pinned
        bit-exact against the compiled reference) and bounded;
  (iii) TBNAV_RBPF_DF_REFERENCE: the product reproduces the reference's brushfire itself (same containers, same
        insert / erase history, same copies on resampling): distance field BIT-EXACT, likelihoods / weights within
        1e-9, Neff and parent lists identical — on the reference's own launch configuration
        (bmapping/launch/slam.launch:19-42: 40 particles, k = 50, 80 x 80 @ 0.05 m) and with a forced resample.
"""
import numpy as np
import pytest

import oracle_api as orc
import rbpf_cases as rc

pytestmark = pytest.mark.gpu


def _dev(gpu_pkg, df_mode=None, pool_bytes=0, **kw):
    from rtn_amd.rbpf import ParticleFilter, default_params
    return ParticleFilter(default_params(**kw), pool_bytes=pool_bytes, df_mode=df_mode)


def _rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def _free_run(gpu_pkg, df_mode, N, k, map_half, walls, n_scans, inc, seed, force_resample_at=None, start=(0.0, 0.0, 0.0),
              oracle_exact_field=False, oracle_window=None, pool_bytes=0, n_beams=360, empty_at=(), walls_at=None, ref_reach=None, **extra):
    """Oracle filter and device filter side by side, same scans, same draws, nothing injected.  oracle_exact_field: the
    oracle's likelihoods read the exact nearest-obstacle distance (the checker of the device's default mode) instead of
    the reference's brushfire."""
    pf_o = orc.PfAPI(orc.pf_params(N=N, k=k, map_min=-map_half, map_max=map_half, pose0=start, **extra),
                     exact_field=oracle_exact_field, window=oracle_window)
    pf_d = _dev(gpu_pkg, df_mode=df_mode, pool_bytes=pool_bytes, N=N, k=k, map_min=-map_half, map_max=map_half, pose0=start, **extra)
    if ref_reach is not None:
        from rtn_amd import capi
        pf_d.setOption(capi.RBPF_OPT_REF_REACH, ref_reach)
    steps, poses = rc.trajectory(n_scans, inc=inc, start=start)
    rng = np.random.default_rng(seed)
    rows = []
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        scan = orc.room_scan(poses[s], n_beams=n_beams, beam_delta_deg=extra.get("beam_delta_deg", 1.0), walls=walls_at(s) if walls_at else walls, rng=rng)
        if s in empty_at:
            scan[:] = 0.01   # every range below range_min: no valid beam
        normals = orc.normal_stream(900 + s, pf_o.normals_per_scan(True), 0.0, 1.0)
        if force_resample_at == s:
            w = np.full(N, 0.2 / N); w[3] += 0.5; w[N // 2] += 0.3; w /= w.sum()
            pf_o.set_particles(w=w); pf_d.setParticles(w=w)
        tr_o = pf_o.slam(scan, u, cur, prev, True, t_icp, normals)
        st = pf_d.SLAM(scan, u, cur, prev, True, t_icp, normals)
        assert st.status == 0 and tr_o["rc"] == 0
        tr_d = pf_d.trace()
        wo, wd = pf_o.particles()[2], pf_d.particles()[2]
        (pose_d, idx_d) = pf_d.getRobotState()
        pose_o = pf_o.particles()[0][pf_o.best()]
        rows.append(dict(
            eta=float(np.max(np.abs(tr_d["eta"] - tr_o["eta"]) / np.abs(tr_o["eta"]))),
            p_scan=float(np.max(np.abs(tr_d["p_scan"] - tr_o["p_scan"]) / np.abs(tr_o["p_scan"]))),
            w=float(np.max(np.abs(wd - wo) / np.abs(wo))),
            neff=(st.neff, tr_o["neff"]), resampled=(st.resampled, tr_o["resampled"]),
            parents_equal=(not st.resampled) or np.array_equal(tr_d["resample_idx"], tr_o["resample_idx"]),
            best=(idx_d, pf_o.best()), best_xy=float(np.hypot(pose_d[1] - pose_o[1], pose_d[2] - pose_o[2])),
            best_th=float(abs(pose_d[0] - pose_o[0])),
            sampled=float(np.max(np.abs(tr_d["sampled"] - tr_o["sampled"]))), p_pose=_rel(tr_d["p_pose"], tr_o["p_pose"]),
            mu=float(np.max(np.abs(tr_d["mu"] - tr_o["mu"]))), new_pose=float(np.max(np.abs(tr_d["new_pose"] - tr_o["new_pose"]))),
            w_raw=_rel(tr_d["weight_raw"], tr_o["weight_raw"])))
    return pf_o, pf_d, rows


def _assert_every_stage(rows, tol=1e-9):
    for s, r in enumerate(rows):
        assert r["sampled"] <= 1e-10 and r["mu"] <= 1e-10 and r["new_pose"] <= 1e-10, (s, r)
        assert max(r["p_scan"], r["p_pose"], r["eta"], r["w_raw"], r["w"]) <= tol, (s, r)                  # north star: 1e-5
        assert r["neff"][0] == r["neff"][1] and r["resampled"][0] == r["resampled"][1] and r["parents_equal"], (s, r)
        assert r["best"][0] == r["best"][1] and r["best_xy"] <= 1e-9 and r["best_th"] <= 1e-9, (s, r)


def test_reference_mode_shipped_config_is_the_reference_end_to_end(gpu_pkg):
    """40 particles, k = 50, 80 x 80, 6 scans, a forced resample after the third, nothing injected."""
    N = 40
    pf_o, pf_d, rows = _free_run(gpu_pkg, "reference", N=N, k=50, map_half=2.0, walls=rc.ROOM_SMALL, n_scans=6,
                                 inc=(0.04, 0.03, 0.02), seed=3, force_resample_at=3)
    assert rows[3]["resampled"] == (1, 1)
    for s, r in enumerate(rows):
        assert r["p_scan"] <= 1e-9 and r["eta"] <= 1e-9 and r["w"] <= 1e-9, (s, r)   # north star: 1e-5
        assert r["neff"][0] == r["neff"][1] and r["resampled"][0] == r["resampled"][1] and r["parents_equal"], (s, r)
        assert r["best"][0] == r["best"][1] and r["best_xy"] <= 1e-9 and r["best_th"] <= 1e-9, (s, r)
    # the field itself: bit-exact, including cells the brushfire gets "wrong" and stale ones
    n_nonexact = 0
    for p in range(N):
        g = pf_o.grid(p).dump()
        assert np.array_equal(pf_d.occDist(p), g["occ_dist"]), p
        assert np.array_equal(pf_d.logOdds(p), g["log_odds"]), p
        occ = np.zeros(pf_d.G, dtype=np.uint8); occ[pf_o.grid(p).occ_cells()] = 1
        exact = orc.exact_edt_codes(occ.reshape(pf_d.xsize, pf_d.xsize), 200, np.full((pf_d.xsize, pf_d.xsize), 0xFFFF, np.uint16))
        n_nonexact += int((pf_d.distCode(p).reshape(pf_d.xsize, pf_d.xsize) != exact).sum())
    assert n_nonexact > 0  # ... i.e. this really is the brushfire, not the exact transform
    pf_d.close()


@pytest.mark.parametrize("mode", ["reference", "query"])
def test_un_injected_run_with_empty_scans(gpu_pkg, mode):
    """A scan without a valid beam as the first scan (empty maps, nothing occupied: the reference's brushfire returns at once,
    grid_mapper.cpp:338) and in the middle of a run, right after a forced resampling (no set operation on freshly copied states) —
    nothing injected, both distance-field modes against their oracle."""
    N = 24
    pf_o, pf_d, rows = _free_run(gpu_pkg, mode, N=N, k=20, map_half=2.0, walls=rc.ROOM_SMALL, n_scans=7, inc=(0.04, 0.03, 0.02), seed=5,
                                 force_resample_at=3, empty_at=(0, 4), oracle_exact_field=(mode == "query"))
    assert rows[3]["resampled"] == (1, 1)
    _assert_every_stage(rows)
    for p in range(N):
        g = pf_o.grid(p).dump()
        assert np.array_equal(pf_d.logOdds(p), g["log_odds"]), p
        if mode == "reference":
            assert np.array_equal(pf_d.occDist(p), g["occ_dist"]), p
    pf_d.close()


def test_reference_mode_shares_one_brushfire_among_particles_in_the_same_state(gpu_pkg):
    """The reference's field is a function of (occupied set with its history, stale field, the scan's insert / erase sequence), and
    the product runs ONE brushfire per distinct such triple instead of one per particle (csrc/ref_field.hpp).  With the shipped
    sampling spread (1e-8 m) and a robot that does not sit on a cell corner, most particles see the same cells change — fewer
    brushfires than particle-scans — and a resample makes copies; every particle's field must still be the oracle's own
    per-particle brushfire bit for bit, stale cells and all."""
    N, n_scans = 48, 6
    pf_o, pf_d, rows = _free_run(gpu_pkg, "reference", N=N, k=50, map_half=2.0, walls=rc.ROOM_SMALL, n_scans=n_scans,
                                 inc=(0.04, 0.03, 0.02), seed=11, force_resample_at=3, start=(0.3, 0.0137, 0.0211))
    assert rows[3]["resampled"] == (1, 1)
    for s, r in enumerate(rows):
        assert r["p_scan"] <= 1e-9 and r["eta"] <= 1e-9 and r["w"] <= 1e-9, (s, r)
        assert r["neff"][0] == r["neff"][1] and r["parents_equal"] and r["best"][0] == r["best"][1], (s, r)
    for p in range(N):
        g = pf_o.grid(p).dump()
        assert np.array_equal(pf_d.occDist(p), g["occ_dist"]), p
        assert np.array_equal(pf_d.logOdds(p), g["log_odds"]), p
    distinct, last, total = pf_d.referenceFieldCounts()
    assert 1 <= distinct <= N and last <= N
    assert total < N * n_scans // 2, (distinct, last, total)   # (every particle on its own would be N * n_scans = 288)
    pf_d.close()


@pytest.mark.parametrize("reach", [0, 1, 3, 6])
def test_reference_mode_lazy_brushfire_is_the_eager_one_for_every_reach(gpu_pkg, reach):
    """Round 6: the reference-mode brushfire stops `reach` cells out (TBNAV_RBPF_OPT_REF_REACH; 0 = to the end, as before) and is
    resumed when a likelihood lookup lands on a cell it has not written (csrc/ref_field.hpp).  Whatever the reach, every stage of
    every scan equals the oracle (whose brushfire always runs to the end), a forced resampling included, and the whole fields —
    whose export finishes the passes — are the oracle's bit for bit."""
    N = 32
    pf_o, pf_d, rows = _free_run(gpu_pkg, "reference", N=N, k=50, map_half=2.0, walls=rc.ROOM_SMALL, n_scans=7,
                                 inc=(0.04, 0.03, 0.02), seed=17, force_resample_at=3, ref_reach=reach)
    assert rows[3]["resampled"] == (1, 1)
    _assert_every_stage(rows)
    st = pf_d.referenceFieldStats()
    assert st["passes"] > 0 and (st["passes_completed"] == st["passes"]) == (reach == 0), st
    for p in range(N):
        g = pf_o.grid(p).dump()
        assert np.array_equal(pf_d.occDist(p), g["occ_dist"]), p
        assert np.array_equal(pf_d.logOdds(p), g["log_odds"]), p
    pf_d.close()


@pytest.mark.parametrize("grid", ["120x120", "400x400"])
def test_reference_mode_walls_that_appear_far_from_the_map_resume_the_passes(gpu_pkg, grid, record_property):
    """The lazy brushfire's hard case: after three scans in a small room the scans come from a room with walls 20-30 cells beyond
    everything mapped so far — their end points land on cells no pass has written (kCodePending on the device).  The proposal kernel
    reports them, exactly those states' passes are resumed until the cells are written (tbnav_rbpf_reference_field_stats:
    states_resumed, proposals_rerun), and every stage still equals the oracle, nothing injected.  On the 400 x 400 grid cells farther
    than cell_radius_ from every obstacle keep STALE values (grid_mapper.cpp:311-314): the whole-field export at the end replays the
    lineages whose passes were cut short (lineages_replayed) and equals the oracle's field bit for bit."""
    big = grid == "400x400"
    N, half = (10, 10.0) if big else (24, 3.0)
    small, large = (-1.0, 1.0, -0.8, 0.9), (-2.4, 2.5, -2.2, 2.3)
    pf_o, pf_d, rows = _free_run(gpu_pkg, "reference", N=N, k=20, map_half=half, walls=small, n_scans=7, inc=(0.05, 0.03, 0.02), seed=23,
                                 force_resample_at=5, walls_at=lambda s: small if s < 3 else large, ref_reach=2)
    _assert_every_stage(rows)
    st = pf_d.referenceFieldStats()
    record_property("reference_field_stats", st)
    print(f"\n[lazy brushfire, {grid}] {st}")
    assert st["states_resumed"] > 0 and st["proposals_rerun"] > 0 and st["passes_completed"] < st["passes"], st
    for p in range(N):
        g = pf_o.grid(p).dump()
        assert np.array_equal(pf_d.occDist(p), g["occ_dist"]), p
        assert np.array_equal(pf_d.logOdds(p), g["log_odds"]), p
    st2 = pf_d.referenceFieldStats()
    if big:
        assert st2["lineages_replayed"] > 0, st2   # (cells out of every obstacle's reach: their values come from the replay)
    # ... and the filter goes on from the finished, exact states as if nothing had happened
    steps, poses = rc.trajectory(9, inc=(0.05, 0.03, 0.02))
    rng = np.random.default_rng(99)
    for s in (7, 8):
        prev, cur, t_icp, u = steps[s]
        scan = orc.room_scan(poses[s], walls=large, rng=rng)
        normals = orc.normal_stream(1900 + s, pf_o.normals_per_scan(True), 0.0, 1.0)
        tr_o = pf_o.slam(scan, u, cur, prev, True, t_icp, normals)
        stt = pf_d.SLAM(scan, u, cur, prev, True, t_icp, normals)
        assert stt.status == 0 and tr_o["rc"] == 0
        wo, wd = pf_o.particles()[2], pf_d.particles()[2]
        assert float(np.max(np.abs(wd - wo) / np.abs(wo))) <= 1e-9 and stt.neff == tr_o["neff"]
    pf_d.close()


def test_reference_mode_replay_in_one_call_with_device_noise_equals_scan_by_scan(gpu_pkg):
    """tbnav_rbpf_slam_batch in the reference-field mode (never pipelined there: every scan's field comes from the host) with the
    standard normals drawn on the device, against one tbnav_rbpf_slam call per scan on a second filter with the same seed and against a
    third with the brushfire run to the end every scan (reach 0): poses, weights, Neff, the resampling and every field bit for bit —
    the rooms change in mid-run so that passes are resumed inside the replay."""
    from rtn_amd import capi
    N, n_scans = 24, 8
    small, large = (-1.0, 1.0, -0.8, 0.9), (-2.4, 2.5, -2.2, 2.3)
    steps, poses = rc.trajectory(n_scans, inc=(0.05, 0.03, 0.02))
    rng = np.random.default_rng(41)
    scans = np.stack([orc.room_scan(poses[s], walls=small if s < 3 else large, rng=rng) for s in range(n_scans)])
    odom = np.array([steps[0][0]] + [st_[1] for st_ in steps]); u_all = np.array([st_[3] for st_ in steps]); ticp = np.array([st_[2] for st_ in steps])
    a, b, e = (_dev(gpu_pkg, df_mode="reference", N=N, k=20, map_min=-3.0, map_max=3.0) for _ in range(3))
    e.setOption(capi.RBPF_OPT_REF_REACH, 0)
    for pf in (a, b, e):
        pf.setSeed(99)
    sts_a = a.SLAMBatch(scans[:5], u_all[:5], odom[:6], ticp[:5])
    w = np.full(N, 0.2 / N); w[2] += 0.5; w[N - 3] += 0.3; w /= w.sum()
    for pf in (a, b, e):
        if pf is not a:
            for s in range(5):
                prev, cur, t_icp, u = steps[s]
                pf.SLAM(scans[s], u, cur, prev, True, t_icp, None)
        pf.setParticles(w=w)   # (the sixth scan resamples)
    sts_a += a.SLAMBatch(scans[5:], u_all[5:], odom[5:], ticp[5:])
    sts_b = []
    for s in range(5, n_scans):
        prev, cur, t_icp, u = steps[s]
        sts_b.append(b.SLAM(scans[s], u, cur, prev, True, t_icp, None)); e.SLAM(scans[s], u, cur, prev, True, t_icp, None)
    assert sts_a[5].resampled == 1 and [(x.neff, x.resampled) for x in sts_a[5:]] == [(x.neff, x.resampled) for x in sts_b]
    for x, y in ((a, b), (a, e)):
        (px, vx, wx), (py, vy, wy) = x.particles(), y.particles()
        assert np.array_equal(px, py) and np.array_equal(vx, vy) and np.array_equal(wx, wy)
    assert a.referenceFieldStats()["states_resumed"] > 0 and e.referenceFieldStats()["states_resumed"] == 0
    for p in range(N):
        assert np.array_equal(a.occDist(p), e.occDist(p)) and np.array_equal(b.occDist(p), e.occDist(p)), p
        assert np.array_equal(a.logOdds(p), e.logOdds(p)), p
    for pf in (a, b, e):
        pf.close()


def test_reference_mode_400x400(gpu_pkg):
    """The 400 x 400 map of BASELINE configs[2] with a small ensemble (the brushfire is 16 ms per particle and scan)."""
    pf_o, pf_d, rows = _free_run(gpu_pkg, "reference", N=12, k=20, map_half=10.0, walls=rc.ROOM_SURVEY, n_scans=4,
                                 inc=(0.07, 0.10, 0.05), seed=5)
    for s, r in enumerate(rows):
        assert r["p_scan"] <= 1e-9 and r["w"] <= 1e-9 and r["neff"][0] == r["neff"][1], (s, r)
    for p in (0, 7, 11):
        assert np.array_equal(pf_d.occDist(p), pf_o.grid(p).dump()["occ_dist"]), p
    pf_d.close()


@pytest.mark.parametrize("cfg", ["shipped_40x80x80", "cfg3_grid_200x400x400"])
def test_default_mode_effect_of_the_exact_field_on_weights_is_bounded(gpu_pkg, cfg, record_property):
    """Contract (ii): what the exact distance field (default) changes relative to the reference's brushfire, on a free
    run with identical noise.  Measured values are recorded in the junit properties and in DESIGN.md section 5; the
    bounds are ~3x what was measured so that a regression in the query shows up."""
    if cfg == "shipped_40x80x80":
        args = dict(N=40, k=50, map_half=2.0, walls=rc.ROOM_SMALL, n_scans=5, inc=(0.04, 0.03, 0.02), seed=3)
    else:
        args = dict(N=200, k=50, map_half=10.0, walls=rc.ROOM_SURVEY, n_scans=5, inc=(0.07, 0.10, 0.05), seed=7)
    pf_o, pf_d, rows = _free_run(gpu_pkg, None, **args)
    worst = {key: max(r[key] for r in rows) for key in ("eta", "p_scan", "w", "best_xy", "best_th")}
    neff_gap = max(abs(r["neff"][0] - r["neff"][1]) for r in rows)
    for key, v in worst.items():
        record_property(key, v)
    record_property("neff_gap", neff_gap)
    print(f"\n[{cfg}] exact field vs reference brushfire, free run: " + ", ".join(f"{k_}={v:.3g}" for k_, v in worst.items()) +
          f", neff gap {neff_gap}, neff per scan {[r['neff'] for r in rows]}")
    # the first scan sees empty maps (likelihood 1 on both sides): identical
    assert rows[0]["w"] <= 1e-9
    # measured on MI355X (round 2): shipped config 4e-14 everywhere (every looked-up cell lies within a cell or two of
    # a wall, where the brushfire IS exact); 200 x 400^2: p_scan / eta / weights 7.5e-3, Neff equal, best pose 4e-16
    lik_bound = 1e-6 if cfg == "shipped_40x80x80" else 0.03
    assert worst["p_scan"] <= lik_bound and worst["eta"] <= lik_bound and worst["w"] <= lik_bound
    assert worst["best_xy"] <= 1e-6 and worst["best_th"] <= 1e-6
    assert neff_gap <= 1
    pf_d.close()


def test_reference_mode_against_the_frozen_trace_G_B4(gpu_pkg):
    """The reference's launch configuration replayed from tests/golden/path_rbpf.npz on the HIP path in
    the reference distance-field mode, nothing injected and no oracle filter beside it: likelihoods / eta / weights within
    1e-9 of the committed trace, Neff, the resampling decision, the parent list, the best particle and its map identical."""
    import os
    import zlib
    from golden.make_golden_paths import RBPF_SCENARIO as S
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "path_rbpf.npz"))
    N, k = S["N"], S["k"]
    pf = _dev(gpu_pkg, df_mode="reference", N=N, k=k)
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))  # noqa: E731
    resampled = 0
    for s in range(S["n_scans"]):
        prev, cur, t_icp, u = g[f"b4_odom_{s}"]
        normals = orc.normal_stream(S["normals_seed"] + s, N * (3 * k + 3) + 1, 0.0, 1.0)
        assert np.uint32(zlib.crc32(np.ascontiguousarray(normals).tobytes())) == g[f"b4_normals_crc_{s}"]
        if s == S["force_resample_at"]:
            pf.setParticles(w=g["b4_forced_w"])
        st = pf.SLAM(g[f"b4_scan_{s}"], u, cur, prev, True, t_icp, normals)
        assert st.status == 0
        tr = pf.trace()
        assert np.allclose(tr["sampled"], g[f"b4_sampled_{s}"], rtol=0, atol=1e-10)
        assert rel(tr["p_scan"], g[f"b4_p_scan_{s}"]) <= 1e-9 and rel(tr["p_pose"], g[f"b4_p_pose_{s}"]) <= 1e-9
        assert rel(tr["eta"], g[f"b4_eta_{s}"]) <= 1e-9 and rel(tr["weight_raw"], g[f"b4_weight_raw_{s}"]) <= 1e-9
        assert np.allclose(tr["mu"], g[f"b4_mu_{s}"], rtol=0, atol=1e-10) and np.allclose(tr["new_pose"], g[f"b4_new_pose_{s}"], rtol=0, atol=1e-10)
        assert st.neff == g[f"b4_neff_{s}"] and st.resampled == g[f"b4_resampled_{s}"]
        if st.resampled:
            resampled += 1
            assert np.array_equal(tr["resample_idx"], g[f"b4_parents_{s}"])
        pose, _, w = pf.particles()
        assert np.allclose(pose, g[f"b4_pose_after_{s}"], rtol=0, atol=1e-10) and rel(w, g[f"b4_weight_after_{s}"]) <= 1e-9
    assert resampled >= 1
    (_, best) = pf.getRobotState()
    assert best == g["b4_best"] and np.array_equal(pf.logOdds(int(best)), g["b4_log_odds_best"])
    pf.close()


def test_cfg3_as_written_1000_particles_400x400_reference_mode_against_the_oracle(gpu_pkg):
    """BASELINE configs[2] AS WRITTEN and nothing injected: 1000 particles, k = 50, 360 beams, 400 x 400 @ 0.05 m, three scans
    with a forced resample at the second — the product in its reference-field mode (the per-particle brushfires spread over
    the host's cores) against the oracle's ParticleFilter (OpenMP over particles): sampled poses, p_scan, p_pose, eta, mu, new
    poses, raw and normalised weights <= 1e-9 (north star 1e-5), Neff / resampling decision / parent list / best particle
    identical, log-odds and distance fields of spot particles bit for bit."""
    import time
    N, k = 1000, 50
    orc.lib().orc_set_threads(8)
    try:
        t0 = time.perf_counter()
        pf_o, pf_d, rows = _free_run(gpu_pkg, "reference", N=N, k=k, map_half=10.0, walls=rc.ROOM_SURVEY, n_scans=3,
                                     inc=rc.TRAJ_SURVEY, seed=11, force_resample_at=1)
        print(f"\n[cfg3 reference mode] 3 scans of {N} particles, oracle + device: {time.perf_counter() - t0:.1f} s")
    finally:
        orc.lib().orc_set_threads(1)
    assert rows[1]["resampled"] == (1, 1)
    for s, r in enumerate(rows):
        assert r["p_scan"] <= 1e-9 and r["eta"] <= 1e-9 and r["w"] <= 1e-9, (s, r)
        assert r["neff"][0] == r["neff"][1] and r["resampled"][0] == r["resampled"][1] and r["parents_equal"], (s, r)
        assert r["best"][0] == r["best"][1] and r["best_xy"] <= 1e-9 and r["best_th"] <= 1e-9, (s, r)
    po, pvo, wo = pf_o.particles()
    pd, pvd, wd = pf_d.particles()
    assert np.allclose(pd, po, rtol=1e-10, atol=1e-15) and np.allclose(pvd, pvo, rtol=1e-10, atol=1e-15)
    for p in (0, 3, 499, 500, 998, 999):
        g = pf_o.grid(p).dump()
        assert np.array_equal(pf_d.logOdds(p), g["log_odds"]), p
        assert np.array_equal(pf_d.occDist(p), g["occ_dist"]), p
    pf_d.close()


@pytest.mark.parametrize("room", ["survey", "bench"])
def test_cfg3_as_written_1000_particles_400x400_default_query_mode_against_the_exact_field_oracle(gpu_pkg, room):
    """The mode bench_rbpf.py's headline number is measured in, held against an oracle AT FULL SIZE (round-3 review, weak 3):
    BASELINE configs[2] as written — 1000 particles, k = 50, 360 beams, 400 x 400 @ 0.05 m, three scans with a forced
    resample at the second — the product in its DEFAULT distance mode (exact nearest-obstacle query, nothing stored, nothing
    injected) against the restated filter with its `exact_field` switch on (oracle/rbpf_oracle.cpp Grid::exact_dist: the
    same filter, likelihoods over the exact distance instead of the brushfire; tests/test_oracle_exact_field.py pins the
    switch).  Sampled poses / mu / new poses 1e-10, p_scan, p_pose, eta, raw and normalised weights <= 1e-9, Neff /
    resampling decision / parent list / best particle identical, six spot particles' log-odds bit for bit.
    The mirror image of ..._reference_mode_against_the_oracle above: that one says the reference-field mode IS the reference
    (brushfire and all); this one says the fast mode is exactly "the reference's filter over the exact field".
    Both rooms the bench has a leg for (round-4 review, weak 3: what is benched must be what is tested): SURVEY 8-d's 6 x 5 m room
    (246 valid beams; the map update picks its two-per-CU / 16-bit-cell form) and the bench's headline room and trajectory
    (rbpf_cases.ROOM_BENCH / TRAJ_BENCH: 360 valid beams) — there the benched instantiation itself, rbpf_raycast_box<512, 8, false,
    4>, is what meets the oracle, at N = 1000."""
    import time
    N, k = 1000, 50
    walls, inc = (rc.ROOM_SURVEY, rc.TRAJ_SURVEY) if room == "survey" else (rc.ROOM_BENCH, rc.TRAJ_BENCH)
    orc.lib().orc_set_threads(8)
    try:
        t0 = time.perf_counter()
        pf_o, pf_d, rows = _free_run(gpu_pkg, None, N=N, k=k, map_half=10.0, walls=walls, n_scans=3,
                                     inc=inc, seed=11, force_resample_at=1, oracle_exact_field=True)
        print(f"\n[cfg3 query mode, {room} room] 3 scans of {N} particles, exact-field oracle + device: {time.perf_counter() - t0:.1f} s; "
              f"kernels {pf_d.lastKernelNames()}")
    finally:
        orc.lib().orc_set_threads(1)
    k_propose, k_raycast, _ = pf_d.lastKernelNames()
    assert k_propose == "rbpf_propose<256, false>", k_propose   # (host normals here — the oracle's draws; the bench's device noise runs <256, true>, held against this form in test_rbpf_gpu.py)
    if room == "bench":
        assert k_raycast == "rbpf_raycast_box<512, 8, false, 4>", (k_raycast, pf_d.raycastBoxCells())
    else:
        assert k_raycast == "rbpf_raycast_box<512, 8, true, 8>", (k_raycast, pf_d.raycastBoxCells())   # (the instantiation of bench_rbpf.py's survey_room leg)
    assert rows[1]["resampled"] == (1, 1)
    _assert_every_stage(rows)
    po, pvo, wo = pf_o.particles()
    pd, pvd, wd = pf_d.particles()
    assert np.allclose(pd, po, rtol=1e-10, atol=1e-15) and np.allclose(pvd, pvo, rtol=1e-10, atol=1e-15)
    for p in (0, 3, 499, 500, 998, 999):
        assert np.array_equal(pf_d.logOdds(p), pf_o.grid(p).dump()["log_odds"]), p
    pf_d.close()
    pf_o.close()


def test_configs4_shard_shape_2000_particles_2000x2000_1080_beams_query_mode_against_the_exact_field_oracle(gpu_pkg):
    """The same comparison at the SHAPE of BASELINE configs[4]'s per-GPU shard: 2000 x 2000 cells @ 0.05 m, 1080-beam scans,
    k = 50, N = 2000 particles (the shard is 12 500; the oracle's particles are dense reference maps — 192 MB each at this
    grid — so its cell store is limited to the 240 x 240-cell window round the room, tests/test_oracle_exact_field.py: a
    write outside it fails the run).  Three scans, forced resample at the second; every stage as above."""
    N, k, bd = 2000, 50, 1.0 / 3.0
    orc.lib().orc_set_threads(8)
    try:
        pf_o, pf_d, rows = _free_run(gpu_pkg, None, N=N, k=k, map_half=50.0, walls=rc.ROOM_SURVEY, n_scans=3,
                                     inc=(0.05, 0.04, 0.03), seed=8, force_resample_at=1, oracle_exact_field=True,
                                     oracle_window=(880, 1120, 880, 1120), pool_bytes=8 << 30, n_beams=1080, beam_delta_deg=bd)
    finally:
        orc.lib().orc_set_threads(1)
    assert (pf_d.xsize, pf_d.ysize) == (2000, 2000)
    # the instantiations bench_rbpf.py's configs4_shard_one_gpu leg times (there with device noise: rbpf_propose<512, true>)
    assert pf_d.lastKernelNames()[:2] == ("rbpf_propose<512, false>", "rbpf_raycast_box<1024, 8, false, 8>"), pf_d.lastKernelNames()
    assert rows[1]["resampled"] == (1, 1)
    _assert_every_stage(rows)
    for p in (0, 1, 999, 1000, 1998, 1999):
        assert np.array_equal(pf_d.logOdds(p), pf_o.grid(p).dump()["log_odds"]), p
    pf_d.close()
    pf_o.close()
