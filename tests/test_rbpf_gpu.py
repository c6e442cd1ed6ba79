"""GPU parity tests of the RBPF scan update: HIP (through the C-ABI) vs the oracle.

Contract (SURVEY.md section 7 "parity contract"):
  * log-odds per cell, occupied sets, Neff, the resample decision and parent list: BIT-EXACT;
  * sampled poses, p_scan, p_pose, eta, mu, new pose, weights: <= 1e-5 relative (north star) — asserted
    at 1e-9 or tighter, with the oracle's (reference brushfire) distance field injected before
    every scan, because the device computes an exact EDT instead of the reference's
    order-dependent brushfire (SURVEY.md hard part 1);
  * the device distance field: bit-exact against the exact-EDT restatement; its deviation from the
    reference brushfire is measured and bounded, not asserted equal.
"""
import numpy as np
import pytest

import oracle_api as orc
import rbpf_cases as rc

pytestmark = pytest.mark.gpu

POSE_RTOL = 1e-10
LIK_RTOL = 1e-9


def _dev(gpu_pkg, df_mode=None, pool_bytes=0, **kw):
    from rtn_amd.rbpf import ParticleFilter, default_params
    return ParticleFilter(default_params(**kw), pool_bytes=pool_bytes, df_mode=df_mode)


def _close(a, b, rtol, atol=0.0):
    return np.allclose(a, b, rtol=rtol, atol=atol)


def _inject(pf_o, pf_d):
    for p in range(pf_o.N):
        pf_d.setOccDist(p, pf_o.grid(p).dump()["occ_dist"])


def _compare_scan(pf_o, pf_d, tr_o, st, icp_ok):
    N, k = pf_o.N, pf_o.k
    tr_d = pf_d.trace()
    assert st.status == 0 and tr_o["rc"] == 0
    if icp_ok:
        assert _close(tr_d["sampled"], tr_o["sampled"], POSE_RTOL, 1e-15)
        assert _close(tr_d["p_scan"], tr_o["p_scan"], LIK_RTOL)
        assert _close(tr_d["p_pose"], tr_o["p_pose"], LIK_RTOL, 1e-300)
        assert _close(tr_d["eta"], tr_o["eta"], LIK_RTOL)
        assert _close(tr_d["mu"], tr_o["mu"], POSE_RTOL, 1e-15)
        assert _close(tr_d["sigma"], tr_o["sigma"], 1e-5, 1e-22)
        assert _close(tr_d["new_pose"], tr_o["new_pose"], POSE_RTOL, 1e-15)
    else:
        assert _close(tr_d["p_scan"][:, 0], tr_o["p_scan"][:, 0], LIK_RTOL)
    assert _close(tr_d["weight_raw"], tr_o["weight_raw"], LIK_RTOL)
    # integer outcomes: bit-exact
    assert (st.neff, st.resampled) == (tr_o["neff"], tr_o["resampled"])
    if st.resampled:
        assert np.array_equal(tr_d["resample_idx"], tr_o["resample_idx"])
    assert abs(st.sum_w - tr_o["sum_w"]) <= LIK_RTOL * abs(tr_o["sum_w"])
    po, pvo, wo = pf_o.particles()
    pd, pvd, wd = pf_d.particles()
    assert _close(pd, po, POSE_RTOL, 1e-15) and _close(pvd, pvo, POSE_RTOL, 1e-15) and _close(wd, wo, LIK_RTOL)
    # maps: log-odds bit-exact, occupied sets equal
    nocc = pf_d.occupiedCount()
    for p in range(N):
        g = pf_o.grid(p)
        assert np.array_equal(pf_d.logOdds(p), g.dump()["log_odds"]), f"log-odds differ for particle {p}"
        assert nocc[p] == len(g.occ_cells())


def _run(gpu_pkg, N, k, map_half, walls, n_scans, icp_ok, beam_delta_deg=1.0, inc=(0.04, 0.03, 0.02), seed=3, **extra):
    n_beams = int(round(360 / beam_delta_deg))
    pf_o = orc.PfAPI(orc.pf_params(N=N, k=k, map_min=-map_half, map_max=map_half, beam_delta_deg=beam_delta_deg, **extra))
    pf_d = _dev(gpu_pkg, N=N, k=k, map_min=-map_half, map_max=map_half, beam_delta_deg=beam_delta_deg, **extra)
    assert (pf_d.xsize, pf_d.ysize) == (pf_o.grid(0).xsize, pf_o.grid(0).ysize)
    steps, poses = rc.trajectory(n_scans, inc=inc)
    rng = np.random.default_rng(seed)
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        scan = orc.room_scan(poses[s], n_beams=n_beams, beam_delta_deg=beam_delta_deg, walls=walls, rng=rng)
        normals = orc.normal_stream(100 + s, pf_o.normals_per_scan(icp_ok), 0.0, 1.0)
        _inject(pf_o, pf_d)
        tr_o = pf_o.slam(scan, u, cur, prev, icp_ok, t_icp, normals)
        st = pf_d.SLAM(scan, u, cur, prev, icp_ok, t_icp, normals)
        _compare_scan(pf_o, pf_d, tr_o, st, icp_ok)
    return pf_o, pf_d


def test_shipped_config_40_particles_80x80_icp_ok(gpu_pkg):
    """bmapping/launch/slam.launch:19-42: 40 particles, k=50, 80x80 @ 0.05 m, 360 beams, 5 scans."""
    pf_o, pf_d = _run(gpu_pkg, N=40, k=50, map_half=2.0, walls=rc.ROOM_SMALL, n_scans=5, icp_ok=True)
    # getRobotState / newMap (particle_filter.cpp:255-291, grid_mapper.cpp:185-226)
    (pose, idx) = pf_d.getRobotState()
    assert idx == pf_o.best()
    assert np.array_equal(pf_d.newMap(), pf_o.grid(idx).grid_map())


@pytest.mark.parametrize("spread", [1e-6, 1e-4, 4e-3])
def test_sampling_spread_from_stable_to_all_unstable_beams(gpu_pkg, spread):
    """The proposal kernel evaluates a beam once for all k samples when no sample can move its end point out of
    the centre's cell, and per sample otherwise.  sample_range variances 1e-6 .. 4e-3 (sigma 1 mm .. 6 cm, against
    5 cm cells) take it from "a few beams near cell borders" through the per-pair path to "every beam unstable"
    (the wave-per-sample path); each must still match the oracle's k * Bv evaluation."""
    _run(gpu_pkg, N=24, k=20, map_half=3.0, walls=rc.ROOM_SMALL, n_scans=4, icp_ok=True, sample_range=[spread * 0.1, spread, spread])


def test_sensor_offset_and_rotation(gpu_pkg):
    """Trs != identity (robot -> laser, rigid2d.cpp:214-224): the sensor transform takes both sincos, and the lookup
    tile is centred on the sensor, not the robot."""
    _run(gpu_pkg, N=24, k=20, map_half=3.0, walls=rc.ROOM_SMALL, n_scans=4, icp_ok=True, Trs=[0.3, 0.05, -0.02])


def test_scan_matching_option_matches_the_oracle_and_pulls_a_bad_guess_in(gpu_pkg):
    """SURVEY.md 8-f N1 (an option, not the reference): each particle refines T(pose) * T_icp against its own map by
    hill climbing on GridMapper::likelihoodFieldModel before sampling.  (1) Device and oracle run the same rule on the
    same (injected) distance fields: matched poses, scores and everything downstream agree at the usual tolerances.
    (2) It does its job: T_icp is deliberately off by (1.5 deg, 3 cm, -2 cm) on the last scans and the matched pose
    is closer to the true increment than the guess was."""
    N, k, n_scans = 24, 20, 6
    pf_o = orc.PfAPI(orc.pf_params(N=N, k=k, map_min=-3.0, map_max=3.0))
    pf_d = _dev(gpu_pkg, N=N, k=k, map_min=-3.0, map_max=3.0)
    steps, poses = rc.trajectory(n_scans, inc=(0.04, 0.03, 0.02))
    rng = np.random.default_rng(5)
    off = np.array([np.deg2rad(1.5), 0.03, -0.02])
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        scan = orc.room_scan(poses[s], walls=rc.ROOM_SMALL, rng=rng)
        normals = orc.normal_stream(700 + s, pf_o.normals_per_scan(True), 0.0, 1.0)
        matching = s >= 3                       # three plain scans build the maps, then the matcher runs on bad guesses
        pf_o.set_scan_matching(matching); pf_d.setScanMatching(matching)
        guess = np.asarray(t_icp) + (off if matching else 0.0)
        _inject(pf_o, pf_d)
        pose_before = pf_o.particles()[0].copy()
        tr_o = pf_o.slam(scan, u, cur, prev, True, guess, normals)
        st = pf_d.SLAM(scan, u, cur, prev, True, guess, normals)
        if matching:
            c_o, sc_o = pf_o.scan_match_result()
            c_d, sc_d = pf_d.scanMatch()
            assert _close(c_d, c_o, POSE_RTOL, 1e-14) and _close(sc_d, sc_o, LIK_RTOL)
            # the true mode is T(pose) * t_icp; the guess was `off` away from it, the match must be nearer
            th, x, y = pose_before[:, 0], pose_before[:, 1], pose_before[:, 2]
            true_xy = np.stack([np.cos(th) * t_icp[1] - np.sin(th) * t_icp[2] + x, np.sin(th) * t_icp[1] + np.cos(th) * t_icp[2] + y], 1)
            guess_xy = np.stack([np.cos(th) * guess[1] - np.sin(th) * guess[2] + x, np.sin(th) * guess[1] + np.cos(th) * guess[2] + y], 1)
            err_match = np.linalg.norm(c_d[:, 1:] - true_xy, axis=1)
            err_guess = np.linalg.norm(guess_xy - true_xy, axis=1)
            assert np.median(err_match) < 0.6 * np.median(err_guess), (np.median(err_match), np.median(err_guess))
        _compare_scan(pf_o, pf_d, tr_o, st, True)


def test_icp_failure_branch_motion_model(gpu_pkg):
    """ICP failed (particle_filter.cpp:161-176): odometry motion-model sample, weight *= scan likelihood."""
    _run(gpu_pkg, N=64, k=10, map_half=2.0, walls=rc.ROOM_SMALL, n_scans=4, icp_ok=False)


def test_400x400_map_and_1080_beams(gpu_pkg):
    """BASELINE configs[2] map (400x400) and configs[4] beam count (1080 @ 1/3 degree)."""
    _run(gpu_pkg, N=6, k=12, map_half=10.0, walls=rc.ROOM_SURVEY, n_scans=3, icp_ok=True, inc=(0.07, 0.10, 0.05))
    _run(gpu_pkg, N=4, k=8, map_half=10.0, walls=rc.ROOM_SURVEY, n_scans=2, icp_ok=True, beam_delta_deg=1.0 / 3.0,
         inc=(0.07, 0.10, 0.05))
    # 1080 beams x 120 samples: the proposal kernel's LDS (scan + per-sample data + bitmap slice) exceeds the 64 KB
    # default dynamic limit
    _run(gpu_pkg, N=3, k=120, map_half=10.0, walls=rc.ROOM_SURVEY, n_scans=2, icp_ok=True, beam_delta_deg=1.0 / 3.0,
         inc=(0.07, 0.10, 0.05))


def test_large_map_2000x2000_with_1080_beams(gpu_pkg):
    """BASELINE configs[4]'s grid (2000 x 2000 @ 0.05 m) and beam count on one GPU (a handful of particles; the config's
    100 k particles need sparse maps and 8 GPUs).  Maps this size have no LDS distance transform: lookups are by query
    only, everything else is the same code path — and must meet the oracle like any other size."""
    _run(gpu_pkg, N=2, k=6, map_half=50.0, walls=rc.ROOM_SURVEY, n_scans=2, icp_ok=True, beam_delta_deg=1.0 / 3.0,
         inc=(0.07, 0.10, 0.05))


def test_on_demand_field_of_a_large_map(gpu_pkg):
    """704 x 704: just past what the LDS transform holds.  get_dist_code then produces the field cell by cell with the
    lookup query; it must equal the brute-force exact EDT (radius cut-off, out-of-reach cells keep their code)."""
    pf_d = _dev(gpu_pkg, N=2, k=3, map_min=-17.6, map_max=17.6)
    xs = pf_d.xsize
    assert xs == 704
    rng = np.random.default_rng(9)
    l_occ = np.log(0.9 / (1 - 0.9))
    for p in range(2):
        occ = np.zeros((xs, xs), dtype=np.uint8)
        if p == 0:
            for r in rng.choice(xs, size=150, replace=False):
                occ[r, rng.choice(xs, size=int(rng.integers(1, 4)), replace=False)] = 1
        else:
            occ[3, 700] = 1; occ[650, 2] = 1            # two far corners: most of the map is out of reach of both
        prev = pf_d.distCode(p).reshape(xs, xs).copy()   # all 0xFFFF
        pf_d.setLogOdds(p, occ.reshape(-1) * l_occ)
        want = orc.exact_edt_codes(occ, 200, prev)
        assert np.array_equal(pf_d.distCode(p).reshape(xs, xs), want)


def test_gated_beams_and_ragged_scan(gpu_pkg):
    """Ranges below range_min / at or above range_max are skipped but still advance the beam angle
    (sensor_model.cpp:77-108); a scan with 357 beams (not a multiple of the wave)."""
    N, k = 5, 7
    pf_o = orc.PfAPI(orc.pf_params(N=N, k=k)); pf_d = _dev(gpu_pkg, N=N, k=k)
    steps, poses = rc.trajectory(2, inc=(0.03, 0.02, 0.02))
    rng = np.random.default_rng(1)
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        scan = orc.room_scan(poses[s], walls=rc.ROOM_SMALL, rng=rng)[:357]
        scan[5:40] = 0.05; scan[100:130] = 3.5; scan[200] = np.inf; scan[201] = np.nan
        normals = orc.normal_stream(7 + s, pf_o.normals_per_scan(True), 0.0, 1.0)
        _inject(pf_o, pf_d)
        tr_o = pf_o.slam(scan, u, cur, prev, True, t_icp, normals)
        st = pf_d.SLAM(scan, u, cur, prev, True, t_icp, normals)
        assert st.n_valid_beams == 357 - 35 - 30 - 2
        _compare_scan(pf_o, pf_d, tr_o, st, True)


@pytest.mark.parametrize("icp_ok", [True, False])
@pytest.mark.parametrize("kind", ["no_valid_beam", "one_valid_beam", "first_scan_empty"])
def test_scans_with_no_or_one_valid_beam(gpu_pkg, kind, icp_ok):
    """The edge of a ragged scan: every range gated out (laserEndPoints returns nothing: the likelihood is an empty product, the map is
    not touched — sensor_model.cpp:77-108, grid_mapper.cpp:100-121, 140-182), a single surviving beam, and an empty scan as the very
    first one (empty maps, no tile allocated yet) — then ordinary scans on top, against the oracle as every other case."""
    N, k = 6, 9
    pf_o = orc.PfAPI(orc.pf_params(N=N, k=k)); pf_d = _dev(gpu_pkg, N=N, k=k)
    steps, poses = rc.trajectory(4, inc=(0.03, 0.02, 0.02))
    rng = np.random.default_rng(2)
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        scan = orc.room_scan(poses[s], walls=rc.ROOM_SMALL, rng=rng)
        want_valid = None
        if (kind == "first_scan_empty" and s == 0) or (kind != "first_scan_empty" and s == 2):
            keep = scan[77]
            scan[:] = np.where(np.arange(scan.size) % 2 == 0, 0.01, 9.0)   # below range_min / beyond range_max
            want_valid = 0
            if kind == "one_valid_beam":
                scan[77] = keep; want_valid = 1
        normals = orc.normal_stream(40 + s, pf_o.normals_per_scan(icp_ok), 0.0, 1.0)
        _inject(pf_o, pf_d)
        tr_o = pf_o.slam(scan, u, cur, prev, icp_ok, t_icp, normals)
        st = pf_d.SLAM(scan, u, cur, prev, icp_ok, t_icp, normals)
        if want_valid is not None:
            assert st.n_valid_beams == want_valid
        _compare_scan(pf_o, pf_d, tr_o, st, icp_ok)
    # ... and with device noise through the batch entry point the empty scan is just another scan
    scans = np.stack([orc.room_scan(poses[s], walls=rc.ROOM_SMALL, rng=rng) for s in range(4)])
    scans[1, :] = 0.01
    odom = np.array([steps[0][0]] + [st_[1] for st_ in steps], dtype=np.float64)
    pf_b = _dev(gpu_pkg, N=N, k=k)
    pf_b.setSeed(9)
    out = pf_b.SLAMBatch(scans, np.array([st_[3] for st_ in steps]), odom, np.array([st_[2] for st_ in steps]), icp_ok=np.full(4, int(icp_ok), dtype=np.int32))
    assert [x.status for x in out] == [0, 0, 0, 0] and out[1].n_valid_beams == 0
    pf_o.close(); pf_d.close(); pf_b.close()


def test_forced_resampling_parents_and_map_copies(gpu_pkg):
    """Skewed weights make Neff < N/2: the parent list is bit-exact (negative Gaussian offset,
    1/(N-1) spacing, clamp at N-1 — particle_filter.cpp:468-500), maps follow their parents and the
    weights are NOT reset."""
    N, k = 16, 6
    for z_last in (-1.3, 0.9):
        pf_o = orc.PfAPI(orc.pf_params(N=N, k=k)); pf_d = _dev(gpu_pkg, N=N, k=k)
        steps, poses = rc.trajectory(3, inc=(0.03, 0.02, 0.02))
        rng = np.random.default_rng(5)
        for s, (prev, cur, t_icp, u) in enumerate(steps):
            scan = orc.room_scan(poses[s], walls=rc.ROOM_SMALL, rng=rng)
            normals = orc.normal_stream(50 + s, pf_o.normals_per_scan(True), 0.0, 1.0)
            if s == 1:
                w = np.full(N, 0.01); w[3] = 0.6; w[11] = 0.25; w /= w.sum()
                pf_o.set_particles(w=w); pf_d.setParticles(w=w)
                normals[-1] = z_last
            _inject(pf_o, pf_d)
            tr_o = pf_o.slam(scan, u, cur, prev, True, t_icp, normals)
            st = pf_d.SLAM(scan, u, cur, prev, True, t_icp, normals)
            _compare_scan(pf_o, pf_d, tr_o, st, True)
            if s == 1:
                assert st.resampled == 1 and len(set(tr_o["resample_idx"].tolist())) < N


def test_distance_field_is_the_exact_edt(gpu_pkg):
    """The device distance field vs the brute-force exact EDT restatement: codes bit-exact, including
    'cells farther than cell_radius keep their previous value' (here: the 0xFFFF initial value)."""
    N, k = 3, 4
    pf_o = orc.PfAPI(orc.pf_params(N=N, k=k, map_min=-6.0, map_max=6.0)); pf_d = _dev(gpu_pkg, N=N, k=k, map_min=-6.0, map_max=6.0)
    xs = pf_d.xsize
    steps, poses = rc.trajectory(3, inc=(0.05, 0.06, 0.04))
    rng = np.random.default_rng(2)
    prev_codes = [np.full((xs, xs), 0xFFFF, dtype=np.uint16) for _ in range(N)]
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        scan = orc.room_scan(poses[s], walls=(-1.2, 1.0, -0.9, 1.1), rng=rng)
        normals = orc.normal_stream(9 + s, pf_o.normals_per_scan(True), 0.0, 1.0)
        _inject(pf_o, pf_d)
        for p in range(N):  # injection overwrote the codes: remember what "previous" now is
            prev_codes[p] = pf_d.distCode(p).reshape(xs, xs)
        pf_o.slam(scan, u, cur, prev, True, t_icp, normals)
        st = pf_d.SLAM(scan, u, cur, prev, True, t_icp, normals)
        assert st.status == 0
        for p in range(N):
            occ = np.zeros(xs * xs, dtype=np.uint8); occ[pf_o.grid(p).occ_cells()] = 1
            want = orc.exact_edt_codes(occ.reshape(xs, xs), 200, prev_codes[p])
            got = pf_d.distCode(p).reshape(xs, xs)
            assert np.array_equal(got, want)
            # decoded metres agree with sqrt(code)*res bit-for-bit
            assert np.array_equal(pf_d.occDist(p).reshape(xs, xs)[got != 0xFFFF], np.sqrt(got[got != 0xFFFF].astype(np.float64)) * 0.05)


def test_distance_field_deviation_from_reference_brushfire_is_bounded(gpu_pkg):
    """REPORTED, not equal: the reference's brushfire (grid_mapper.cpp:333-435) is not an exact EDT
    (SURVEY.md hard part 1: ~4 % of cells differ, < 1 cell).  Bound the deviation so a regression in
    either direction shows up."""
    N, k = 2, 4
    pf_o = orc.PfAPI(orc.pf_params(N=N, k=k, map_min=-10.0, map_max=10.0)); pf_d = _dev(gpu_pkg, N=N, k=k, map_min=-10.0, map_max=10.0)
    steps, poses = rc.trajectory(6, inc=(0.07, 0.10, 0.05))
    rng = np.random.default_rng(7)
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        scan = orc.room_scan(poses[s], walls=rc.ROOM_SURVEY, rng=rng)
        normals = orc.normal_stream(20 + s, pf_o.normals_per_scan(True), 0.0, 1.0)
        _inject(pf_o, pf_d)
        pf_o.slam(scan, u, cur, prev, True, t_icp, normals)
        pf_d.SLAM(scan, u, cur, prev, True, t_icp, normals)
    ref = pf_o.grid(0).dump()["occ_dist"]; dev = pf_d.occDist(0)
    reach = (ref != 10.0) & (dev != 10.0)
    diff = np.abs(ref - dev)[reach]
    frac = float(np.mean(diff > 0)); worst = float(diff.max())
    near = reach & (dev <= 4 * 0.05)
    frac_near = float(np.mean(np.abs(ref - dev)[near] > 0))
    print(f"\n[edt-vs-brushfire] cells reached {reach.sum()}, differing {frac*100:.2f} % (within 4 cells of an obstacle: "
          f"{frac_near*100:.2f} % of {near.sum()}), max |delta| {worst:.4f} m")
    assert np.all(dev[reach] <= ref[reach] + 1e-12)   # an exact EDT is never farther than the brushfire's path-propagated distance
    assert frac < 0.5 and worst <= 0.05 * 2.0         # measured on MI355X round 1: 27.9 % of cells, max 0.083 m (< 2 cells)


def test_reference_exceptions_become_status_codes(gpu_pkg):
    N, k = 4, 5
    pf_d = _dev(gpu_pkg, N=N, k=k)
    normals = orc.normal_stream(1, pf_d.numNormals(True), 0.0, 1.0)
    far = np.full(360, 3.0, dtype=np.float32)  # end points 3 m out on a +-2 m map: "... NOT in the bounds of the world"
    st = pf_d.SLAM(far, (0, 0.05, 0), (0, 0.05, 0), (0, 0, 0), True, (0, 0.05, 0), normals, check=False)
    assert st.status == gpu_pkg.capi.ERR_OUT_OF_WORLD
    with pytest.raises(gpu_pkg.capi.TbnavError) as ei:
        pf_d.SLAM(far, (0, 0.05, 0), (0, 0.05, 0), (0, 0, 0), True, (0, 0.05, 0), normals)
    assert "NOT in the bounds of the world" in str(ei.value)
    # zero motion with zero sampling noise: pdfNormal's variance is 0 -> the reference throws
    pf2 = _dev(gpu_pkg, N=N, k=k)
    ok_scan = orc.room_scan((0, 0, 0), walls=rc.ROOM_SMALL)
    st = pf2.SLAM(ok_scan, (0, 0, 0), (0, 0, 0), (0, 0, 0), True, (0, 0, 0), np.zeros(pf2.numNormals(True)), check=False)
    assert st.status == gpu_pkg.capi.ERR_PDF_VARIANCE
    # occ_dist values the reference cannot produce are rejected
    with pytest.raises(gpu_pkg.capi.TbnavError):
        pf2.setOccDist(0, np.full(pf2.G, 0.123))


def test_cfg3_full_size_properties(gpu_pkg):
    """BASELINE configs[2]: 1000 particles, 360 beams, 400x400 — size-independent properties:
    weights normalised, every particle's log-odds equal the oracle GridMapper driven with that
    particle's own pose (3 spot checks), distance field idempotent under an empty scan, resample
    global helper agrees with the device's decision."""
    from rtn_amd.rbpf import resample_global
    N, k = 1000, 50
    pf_d = _dev(gpu_pkg, N=N, k=k, map_min=-10.0, map_max=10.0)
    pf_d.setTiming(True)
    steps, poses = rc.trajectory(3, inc=(0.07, 0.10, 0.05))
    rng = np.random.default_rng(7)
    grids = {p: orc.GridAPI("orc", grid=(0.05, -10.0, 10.0, -10.0, 10.0)) for p in (0, 499, 999)}
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        scan = orc.room_scan(poses[s], walls=rc.ROOM_SURVEY, rng=rng)
        normals = orc.normal_stream(300 + s, pf_d.numNormals(True), 0.0, 1.0)
        st = pf_d.SLAM(scan, u, cur, prev, True, t_icp, normals)
        assert st.status == 0 and st.resampled == 0
        pose, _, w = pf_d.particles()
        assert abs(w.sum() - 1.0) < 1e-12 and st.neff == int(1.0 / np.sum(w * w))
        raw = pf_d.trace()["weight_raw"]
        parents, wn, st2 = resample_global(raw, normals[-1])
        assert (st2.neff, st2.resampled) == (st.neff, st.resampled) and np.array_equal(wn, w)
        for p, g in grids.items():
            assert g.integrate_scan(scan, pose[p], esdf=False) == 0
            assert np.array_equal(pf_d.logOdds(p), g.dump()["log_odds"])
    print("\n[cfg3 kernel ms]", pf_d.kernelMs())
    codes = pf_d.distCode(0).copy()
    empty = np.full(360, 9.0, dtype=np.float32)  # all beams gated out: maps untouched
    normals = orc.normal_stream(1, pf_d.numNormals(True), 0.0, 1.0)
    prev, cur, t_icp, u = steps[-1]
    pf_d.SLAM(empty, u, cur + 0.01, cur, True, t_icp, normals)
    assert np.array_equal(pf_d.distCode(0), codes)


def test_distance_lookup_modes_are_bit_identical(gpu_pkg):
    """The scan likelihood gets its distance codes three ways: whole-map transform after every
    update ("full", the reference's data flow), a per-particle window refreshed before the update ("window"), or
    an exact nearest-obstacle query on the occupancy bitmap at each lookup ("query", the default — no transform in
    the SLAM path).  All three must give the same bits everywhere: sampled-pose likelihoods, weights, poses, maps,
    and the field that get_dist_code materialises afterwards."""
    N, k, n_scans = 96, 20, 6
    steps, poses = rc.trajectory(n_scans, inc=(0.07, 0.10, 0.05))
    rng = np.random.default_rng(11)
    scans = [orc.room_scan(poses[s], walls=rc.ROOM_SURVEY, rng=rng) for s in range(n_scans)]
    outs = {}
    for mode in ("full", "window", "query"):
        pf = _dev(gpu_pkg, df_mode=mode, N=N, k=k, map_min=-10.0, map_max=10.0)
        rec = []
        for s, (prev, cur, t_icp, u) in enumerate(steps):
            normals = orc.normal_stream(500 + s, pf.numNormals(True), 0.0, 1.0)
            st = pf.SLAM(scans[s], u, cur, prev, True, t_icp, normals)
            assert st.status == 0
            tr = pf.trace()
            rec.append((tr["p_scan"].copy(), tr["weight_raw"].copy(), tr["new_pose"].copy(), st.neff, st.resampled))
        pose, prevp, w = pf.particles()
        outs[mode] = (rec, pose, w, pf.logOdds(5).copy(), pf.distCode(5).copy(), pf.distCode(N - 1).copy())
        pf.close()
    for mode in ("window", "query"):
        for a, b in zip(outs["full"][0], outs[mode][0]):
            assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])) and a[3:] == b[3:], mode
        for x, y in zip(outs["full"][1:], outs[mode][1:]):
            assert np.array_equal(x, y), mode
    assert (outs["full"][4] != 0xFFFF).sum() > 1000      # the field really is populated


@pytest.mark.parametrize("rows_used,expect", [(140, "compact-144"), (280, "compact-288"), (400, "general")])
def test_distance_field_tiers_random_occupancy(gpu_pkg, rows_used, expect):
    """The distance transform picks its kernel per particle from the number of non-empty map rows
    (<= 144, <= 288, any).  Random occupancy patterns exercise each tier, ragged column tiles
    (400 = 6*64 + 16) and the radius cut-off; codes must equal the brute-force exact EDT."""
    N, k = 2, 3
    pf_d = _dev(gpu_pkg, N=N, k=k, map_min=-10.0, map_max=10.0)
    xs = pf_d.xsize
    rng = np.random.default_rng(rows_used)
    l_occ = np.log(0.9 / (1 - 0.9))
    occs = []
    for p in range(N):
        occ = np.zeros((xs, xs), dtype=np.uint8)
        rows = rng.choice(xs, size=rows_used, replace=False)
        for r in rows:
            occ[r, rng.choice(xs, size=int(rng.integers(1, 4)), replace=False)] = 1
        if p == 1 and rows_used == 140:
            occ[:] = 0; occ[5, 7] = 1   # single obstacle in a corner: most of the map is beyond the 200-cell radius
        pf_d.setLogOdds(p, occ.reshape(-1) * l_occ)
        occs.append(occ)
    prev_codes = [pf_d.distCode(p).reshape(xs, xs) for p in range(N)]
    empty = np.full(360, 9.0, dtype=np.float32)
    normals = orc.normal_stream(1, pf_d.numNormals(True), 0.0, 1.0)
    st = pf_d.SLAM(empty, (0.05, 0.1, 0), (0.05, 0.1, 0.02), (0, 0, 0), True, (0.05, 0.1, 0.02), normals)
    assert st.status == 0
    assert pf_d.occupiedCount().tolist() == [int(o.sum()) for o in occs]
    for p in range(N):
        want = orc.exact_edt_codes(occs[p], 200, prev_codes[p])
        assert np.array_equal(pf_d.distCode(p).reshape(xs, xs), want), expect


def test_device_noise_source_statistics_and_filter_health(gpu_pkg):
    """normals == NULL: the standard normals are drawn on the device (Philox + Box-Muller).  Moments of
    the stream, determinism in (seed, scan count), and the filter it drives stays healthy: poses track the
    trajectory like the host-noise run, weights normalised, maps populated."""
    N, k = 256, 50
    pf = _dev(gpu_pkg, N=N, k=k)
    pf.setSeed(99)
    steps, poses = rc.trajectory(4, inc=(0.04, 0.03, 0.02))
    rng = np.random.default_rng(3)
    scans = [orc.room_scan(poses[s], walls=rc.ROOM_SMALL, rng=rng) for s in range(4)]
    streams = []
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        st = pf.SLAM(scans[s], u, cur, prev, True, t_icp, None)
        assert st.status == 0
        streams.append(pf.lastNormals(pf.numNormals(True)))
    z = np.concatenate(streams)
    n = z.size
    assert abs(z.mean()) < 5 / np.sqrt(n) and abs(z.var() - 1.0) < 0.02
    assert abs(np.mean(z ** 3)) < 0.05 and abs(np.mean(z ** 4) - 3.0) < 0.1
    assert abs(np.corrcoef(z[0::2], z[1::2])[0, 1]) < 0.01       # the two Box-Muller outputs are uncorrelated
    assert not np.array_equal(streams[0], streams[1])
    pose, _, w = pf.particles()
    assert abs(w.sum() - 1.0) < 1e-12
    # same inputs through the oracle filter with host normals: the two particle clouds coincide (the sampling
    # spread is 1e-4 m, so "coincide" = means within 1 mm, spread within 1 mm)
    pf_o = orc.PfAPI(orc.pf_params(N=64, k=k))
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        pf_o.slam(scans[s], u, cur, prev, True, t_icp, orc.normal_stream(7 + s, pf_o.normals_per_scan(True), 0.0, 1.0), trace=False)
    po, _, _ = pf_o.particles()
    assert np.allclose(pose.mean(0), po.mean(0), atol=1e-3) and np.all(pose.std(0) < 1e-3)
    assert pf.occupiedCount().min() > 100
    # same seed -> same stream
    pf2 = _dev(gpu_pkg, N=N, k=k); pf2.setSeed(99)
    prev, cur, t_icp, u = steps[0]
    pf2.SLAM(scans[0], u, cur, prev, True, t_icp, None)
    assert np.array_equal(pf2.lastNormals(pf2.numNormals(True)), streams[0])


@pytest.mark.parametrize("N,k,icp", [(64, 50, "ok"), (40, 300, "ok"), (33, 7, "alternating"), (1000, 50, "ok")])
def test_noise_drawn_inside_the_proposal_kernel_equals_sampled_noise(gpu_pkg, N, k, icp):
    """Round 5: with device noise the standard normals can be drawn INSIDE rbpf_propose (TBNAV_RBPF_OPT_NOISE_IN_KERNEL = 1: nothing
    stored; the beam table rides in through the launch's leading workgroup; since round 6 the option, not the default — the stored-first
    kernel is the faster one).  Same contract as MPPI's in-kernel sampler: the filter equals, bit for bit, (a) the
    filter with TBNAV_RBPF_OPT_NOISE_IN_KERNEL = 0 (rbpf_sample_normals stores the stream first, the default) and (b) a
    filter fed the regenerated stream (tbnav_rbpf_get_normals) as HOST normals — the path the oracle tests drive.  k = 300: more
    samples than threads; alternating: ICP-failed scans (three normals per particle); a forced resampling on the way (the
    resampling offset's normal is the one value the kernel does store)."""
    from rtn_amd import capi
    a, b, h = _dev(gpu_pkg, N=N, k=k), _dev(gpu_pkg, N=N, k=k), _dev(gpu_pkg, N=N, k=k)
    a.setOption(capi.RBPF_OPT_NOISE_IN_KERNEL, 1)
    for pf in (a, b, h):
        pf.setSeed(4242)
    n_scans = 5
    steps, poses = rc.trajectory(n_scans, inc=(0.04, 0.03, 0.02))
    rng = np.random.default_rng(3)
    scans = [orc.room_scan(poses[s], walls=rc.ROOM_SMALL, rng=rng) for s in range(n_scans)]
    resampled = 0
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        ok = True if icp == "ok" else (s % 2 == 0)
        if s == 2:
            w = np.full(N, 0.2 / N); w[1] += 0.5; w[N - 2] += 0.3; w /= w.sum()
            for pf in (a, b, h):
                pf.setParticles(w=w)
        sa = a.SLAM(scans[s], u, cur, prev, ok, t_icp, None)
        sb = b.SLAM(scans[s], u, cur, prev, ok, t_icp, None)
        assert sa.status == 0 and sb.status == 0
        za, zb = a.lastNormals(a.numNormals(ok)), b.lastNormals(b.numNormals(ok))
        assert np.array_equal(za, zb)                                   # regenerated from the counters == stored by the sample kernel
        sh = h.SLAM(scans[s], u, cur, prev, ok, t_icp, za)             # ... and fed back as host normals
        assert (sa.neff, sa.resampled) == (sb.neff, sb.resampled) == (sh.neff, sh.resampled)
        resampled += int(sa.resampled)
        for x, y in ((a, b), (a, h)):
            (px, vx, wx), (py, vy, wy) = x.particles(), y.particles()
            assert np.array_equal(px, py) and np.array_equal(vx, vy) and np.array_equal(wx, wy), s
    assert resampled >= 1
    for p in (0, N // 2, N - 1):
        assert np.array_equal(a.logOdds(p), b.logOdds(p)) and np.array_equal(a.logOdds(p), h.logOdds(p))
    ka, kb, kh = a.lastKernelNames()[0], b.lastKernelNames()[0], h.lastKernelNames()[0]
    assert ka.endswith(", true>") and kb == kh == ka.replace(", true>", ", false>"), (ka, kb, kh)   # the two instantiations of rbpf_propose<NT, DN>
    for pf in (a, b, h):
        pf.close()


def test_device_map_export_matches_glibc_evaluation(gpu_pkg):
    """newMap on the device: int8 {-1, 0, 100, (int8)(prob*100)}, transposed
    (grid_mapper.cpp:185-226).  The device never evaluates prob; it uses log-odds break points found with glibc
    at create time.  Checked on every log-odds value reachable with up to 8 hits per cell in any order, the
    knife edges (one occupied hit = exactly 0.9, one free hit = exactly 0.35, untouched = exactly 0.5) and
    their neighbours in the last bit; the expected value is computed with the oracle's glibc logOdds2Prob."""
    r = orc.RigidAPI("orc")
    l_occ, l_free = r.prob_to_log_odds(0.90), r.prob_to_log_odds(0.35)
    vals = {0.0}
    frontier = {0.0}
    for _ in range(8):                       # every add ORDER gives its own rounding: enumerate sequences
        frontier = {v + d for v in frontier for d in (l_occ, l_free)}
        vals |= frontier
    edge = [l_occ, l_free, 0.0, l_occ + l_free]
    for e in edge:
        for _ in range(3):
            vals |= {np.nextafter(e, 9.0), np.nextafter(e, -9.0)}
            e = np.nextafter(e, 9.0)
    vals |= {1e-17, -1e-17, 2.3e-16, -2.3e-16, 1e-15, -1e-15, 5.0, -5.0, 40.0, -40.0}
    vals = np.array(sorted(vals))
    pf = _dev(gpu_pkg, N=3, k=3)             # 80 x 80 = 6400 cells
    assert vals.size <= pf.G
    lo = np.zeros(pf.G); lo[:vals.size] = vals
    pf.setLogOdds(1, lo)
    pf.setParticles(w=np.array([0.2, 0.5, 0.3]))   # particle 1 is the arg-max
    (pose, idx) = pf.getRobotState()
    assert idx == 1
    got = pf.newMap().reshape(pf.xsize, pf.xsize).T.reshape(-1)   # undo the transpose
    def expect(l):
        p = r.log_odds_to_prob(l)
        if p == 0.5: return -1
        if p >= 0.90: return 100
        if p <= 0.35: return 0
        return int(np.int8(int(p * 100)))
    want = np.array([expect(l) for l in lo[:vals.size]])
    bad = np.flatnonzero(got[:vals.size] != want)
    assert bad.size == 0, (vals[bad][:5], got[bad][:5], want[bad][:5])
    assert np.all(got[vals.size:] == -1)
    assert set(np.unique(want)) >= {-1, 0, 100, 82}   # occ then free -> 0.829 -> 82
    # arg-max tie rule: strict '>', first wins; all-zero weights -> index 0
    pf.setParticles(w=np.array([0.4, 0.4, 0.2])); assert pf.getRobotState()[1] == 0
    pf.setParticles(w=np.array([0.0, 0.0, 0.0])); assert pf.getRobotState()[1] == 0


def test_batch_replay_equals_one_call_per_scan(gpu_pkg):
    """tbnav_rbpf_slam_batch is n tbnav_rbpf_slam calls made from C: same seed, same scans -> the same filter, bit for
    bit (poses, weights, every map), and the same per-scan stats."""
    N, k, n_scans = 64, 10, 6
    steps, poses = rc.trajectory(n_scans, inc=(0.05, 0.04, 0.03))
    rng = np.random.default_rng(21)
    scans = np.stack([orc.room_scan(poses[s], walls=rc.ROOM_SMALL, rng=rng) for s in range(n_scans)])
    odom = np.array([steps[0][0]] + [st[1] for st in steps], dtype=np.float64)
    u = np.array([st[3] for st in steps], dtype=np.float64)
    t_icp = np.array([st[2] for st in steps], dtype=np.float64)
    a = _dev(gpu_pkg, N=N, k=k)
    b = _dev(gpu_pkg, N=N, k=k)
    a.setSeed(77); b.setSeed(77)
    one = [a.SLAM(scans[s], steps[s][3], steps[s][1], steps[s][0], True, steps[s][2], None) for s in range(n_scans)]
    many = b.SLAMBatch(scans[:2], u[:2], odom[:3], t_icp[:2]) + b.SLAMBatch(scans[2:], u[2:], odom[2:], t_icp[2:])
    for x, y in zip(one, many):
        assert (x.status, x.neff, x.resampled, x.n_valid_beams) == (y.status, y.neff, y.resampled, y.n_valid_beams)
        assert x.sum_w == y.sum_w and x.sq_sum == y.sq_sum
    pa, pb = a.particles(), b.particles()
    for q in range(3):
        assert np.array_equal(pa[q], pb[q])
    for m in (0, 17, N - 1):
        assert np.array_equal(a.logOdds(m), b.logOdds(m))
    a.close(); b.close()


def _logged_run(n_scans, seed):
    steps, poses = rc.trajectory(n_scans, inc=(0.05, 0.04, 0.03))
    rng = np.random.default_rng(seed)
    scans = np.stack([orc.room_scan(poses[s], walls=rc.ROOM_SMALL, rng=rng) for s in range(n_scans)])
    odom = np.array([steps[0][0]] + [st[1] for st in steps], dtype=np.float64)
    u = np.array([st[3] for st in steps], dtype=np.float64)
    t_icp = np.array([st[2] for st in steps], dtype=np.float64)
    return steps, scans, odom, u, t_icp


@pytest.mark.gpu
@pytest.mark.parametrize("matcher", [False, True])
def test_pipelined_batch_equals_synchronous_scans_across_resamples(gpu_pkg, matcher):
    """tbnav_rbpf_slam_batch keeps two scans in the stream: scan s + 1 is enqueued before the host knows whether scan s
    resamples, and does nothing on the device if it does.  A run that resamples in the middle must come out bit for bit
    like one synchronous call per scan, and like the batch with the pipeline switched off."""
    N, k, n_scans = 48, 10, 16
    steps, scans_all, odom_all, u_all, t_all = _logged_run(n_scans + 2, 5)
    scans, odom, u, t_icp = scans_all[:n_scans], odom_all[:n_scans + 1], u_all[:n_scans], t_all[:n_scans]
    # wider sampling and a sharper sensor model than the shipped ones: the weights spread and the run resamples by itself
    # (scans 8, 10, 11 and 14 without the matcher — twice in a row among them)
    pfs = [_dev(gpu_pkg, N=N, k=k, sigma_hit=0.02, sample_range=[1e-5, 3e-4, 3e-4]) for _ in range(3)]
    for pf in pfs:
        pf.setSeed(1234)
        if matcher:
            pf.setScanMatching(True)
    pfs[2].setOption(gpu_pkg.capi.RBPF_OPT_BATCH_PIPELINE, 0)
    one = [pfs[0].SLAM(scans[s], steps[s][3], steps[s][1], steps[s][0], True, steps[s][2], None) for s in range(n_scans)]
    piped = pfs[1].SLAMBatch(scans, u, odom, t_icp)
    plain = pfs[2].SLAMBatch(scans, u, odom, t_icp)
    fired = [s for s in range(n_scans) if one[s].resampled]
    assert any(0 < s < n_scans - 1 for s in fired), "the run must resample in the middle for this test to mean anything"
    for other in (piped, plain):
        for x, y in zip(one, other):
            assert (x.status, x.neff, x.resampled, x.n_valid_beams) == (y.status, y.neff, y.resampled, y.n_valid_beams)
            assert x.sum_w == y.sum_w and x.sq_sum == y.sq_sum
    ref = pfs[0].particles()
    for pf in pfs[1:]:
        got = pf.particles()
        for q in range(3):
            assert np.array_equal(ref[q], got[q])
        for m in range(0, N, 5):
            assert np.array_equal(pfs[0].logOdds(m), pf.logOdds(m))
    # and the replay can go on from where the batch stopped
    more = [pf.SLAMBatch(scans_all[n_scans:], u_all[n_scans:], odom_all[n_scans:], t_all[n_scans:]) for pf in pfs[1:]]
    assert [(x.neff, x.sum_w) for x in more[0]] == [(x.neff, x.sum_w) for x in more[1]]
    for pf in pfs:
        pf.close()


@pytest.mark.gpu
def test_pipelined_batch_stops_at_the_first_failing_scan(gpu_pkg):
    N, k, n_scans = 16, 5, 5
    steps, scans, odom, u, t_icp = _logged_run(n_scans, 9)
    scans[2, :] = 3.0  # end points outside the +-2 m world: the reference throws in scan 2
    pf = _dev(gpu_pkg, N=N, k=k)
    pf.setSeed(3)
    out = pf.SLAMBatch(scans, u, odom, t_icp, check=False)
    assert [o.status for o in out[:3]] == [0, 0, gpu_pkg.capi.ERR_OUT_OF_WORLD]
    with pytest.raises(gpu_pkg.capi.TbnavError):
        pf.SLAMBatch(scans, u, odom, t_icp)
    pf.close()


@pytest.mark.parametrize("n", [1, 2, 3, 37, 1000, 2047, 2048, 2049, 5000, 100_000])
def test_parallel_exact_chains_equal_the_sequential_sums_bit_for_bit(gpu_pkg, n):
    """rbpf_normalize's three left-to-right sums (sum of weights, sum of squares of the normalised weights -> Neff, the comb's
    running sum -> parent list: particle_filter.cpp:442-500) are evaluated WITHOUT a chain of dependent adds (chain_exact: integer
    prefix sums per binade, ties and binade crossings by single plain adds).  Against the plain sequential loops on the host
    (tbnav_rbpf_resample_global) — sums bit for bit, Neff, decision, every parent — on weight vectors built to hit what breaks the
    pattern: uniform weights (every add a tie candidate: dyadic values), one dominant weight (crossings up and long runs in one
    binade), 20 orders of magnitude of spread, zeros, weights that are exact multiples of the running sum's half-ulp."""
    import ctypes as C
    import torch
    from rtn_amd import capi
    from rtn_amd.rbpf import resample_global
    pf = _dev(gpu_pkg, N=1, k=2)
    rng = np.random.default_rng(n)
    cases = {
        "uniform": np.full(n, 1.0 / max(n, 1)),
        "dyadic": np.ldexp(rng.integers(1, 1 << 12, n).astype(np.float64), -int(np.ceil(np.log2(max(n, 2))) + 20)),
        "random": rng.random(n),
        "lognormal": np.exp(rng.normal(0.0, 8.0, n)),
        "dominant": np.where(np.arange(n) == n // 3, 1.0, rng.random(n) * 1e-9),
        "zeros": np.where(rng.random(n) < 0.5, 0.0, rng.random(n)),
        "half_ulps": np.ldexp(2.0 * rng.integers(1, 1 << 20, n).astype(np.float64) + 1.0, -53 - 8),   # odd multiples of 2^-61: ties once the sum passes 2^-8
        "skewed_resample": np.where(np.arange(n) % 97 == 5, 1.0, 1e-6 * rng.random(n)),
    }
    for name, w in cases.items():
        if n == 1:
            w = np.array([0.37])
        if w.sum() == 0.0:
            w[0] = 1.0
        z = float(rng.standard_normal())
        parents_h, wn_h, st_h = resample_global(w, z)
        dw = torch.from_numpy(w).cuda()
        parents_d = np.empty(n, dtype=np.int32)
        st = capi.RbpfStats()
        capi.check(pf._L.tbnav_rbpf_resample_global_dev(pf._h, dw.data_ptr(), n, 0, C.c_double(z), parents_d.ctypes.data, C.byref(st)), "resample_global_dev")
        assert (st.sum_w, st.sq_sum, st.neff, st.resampled) == (st_h.sum_w, st_h.sq_sum, st_h.neff, st_h.resampled), (name, n, st.sum_w - st_h.sum_w, st.sq_sum - st_h.sq_sum)
        if st.resampled:
            assert np.array_equal(parents_d, parents_h), (name, n)
    pf.close()


def test_repeated_adds_without_the_chain_equal_the_plain_loop_bit_for_bit(gpu_pkg):
    """A cell that n beams cross takes n times  x = fl(x + l)  (grid_mapper.cpp:438-477: one add per beam) — for the robot's own cell
    n is the number of beams.  add_repeated (rbpf_device.hpp) evaluates that without the chain of dependent adds: integer steps while x
    stays in one binade, plain adds across its ends, at ties and where the signs differ.  Against the plain loop in numpy, bit for
    bit: the shipped log-odds, random addends, powers of two (every add a tie candidate), addends far below / above x, zeros,
    infinities, sign changes, and counts from 0 to 65535."""
    import ctypes as C
    from rtn_amd import capi
    lib = capi.lib()
    rng = np.random.default_rng(5)
    l_free, l_occ = np.log(0.35 / 0.65), np.log(0.9 / 0.1)
    xs, ds, ns = [], [], []
    def add(x, d, n):
        x, d, n = np.broadcast_arrays(np.asarray(x, np.float64), np.asarray(d, np.float64), np.asarray(n, np.int64))
        xs.append(x.ravel().copy()); ds.append(d.ravel().copy()); ns.append(n.ravel().copy())
    m = 3000
    for l in (l_free, l_occ, -l_occ):
        add(-np.abs(rng.normal(0, 300, m)), l, rng.integers(0, 400, m))           # what the maps hold
        add(rng.normal(0, 5, m), l, rng.integers(0, 40, m))                       # round zero: sign changes, small binades
        add(np.ldexp(rng.random(m) + 1.0, rng.integers(-30, 60, m)) * np.sign(l), l, rng.integers(0, 65536, m))
        add(0.0, l, np.arange(0, 600))                                            # a fresh cell: every binade from zero up
    add(rng.normal(0, 100, m), rng.normal(0, 2, m), rng.integers(0, 3000, m))     # any addend
    add(np.ldexp(rng.integers(1, 1 << 20, m).astype(np.float64), -10), np.ldexp(1.0, rng.integers(-14, 3, m)), rng.integers(0, 5000, m))   # ties
    add(np.ldexp(2.0 * rng.integers(1 << 51, 1 << 52, m).astype(np.float64) + 1.0, -40), np.ldexp(1.0, -41), rng.integers(0, 200, m))      # x odd, d = ulp/2
    add(rng.normal(0, 1, m) * 1e15, l_free, rng.integers(0, 65536, m))            # addend near / below half an ulp
    add(rng.normal(0, 1, m) * 1e17, l_free, rng.integers(0, 65536, m))
    add([np.inf, -np.inf, np.nan, 1.0, 0.0, -0.0, 5e-324, -1e-310], [l_free, l_free, l_free, np.inf, 0.0, -0.0, l_free, -1e-310], 1000)
    add(-np.ldexp(1.0, np.arange(1, 40)) + 0.25, l_free, 50)                      # just inside a binade's end
    x = np.concatenate(xs); d = np.concatenate(ds); n = np.concatenate(ns).astype(np.int32)
    out = np.empty_like(x)
    rc = lib.tbnav_rbpf_add_repeated(x.ctypes.data, d.ctypes.data, n.ctypes.data, out.ctypes.data, C.c_int64(len(x)))
    assert rc == 0
    want = x.copy()
    with np.errstate(all="ignore"):
        for step in range(int(n.max())):
            live = n > step
            want[live] = want[live] + d[live]
    same = (out.view(np.uint64) == want.view(np.uint64)) | (np.isnan(out) & np.isnan(want))
    bad = np.flatnonzero(~same)
    assert bad.size == 0, (bad[:5], x[bad[:5]], d[bad[:5]], n[bad[:5]], out[bad[:5]], want[bad[:5]])
