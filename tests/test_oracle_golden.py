"""The oracle restatement vs (a) the golden fixtures generated from the real reference build
(tests/golden/*.npz, made by tests/golden/make_golden.py) — bit-exact — and (b) the known answers
the reference's own gtests hold for the adjacent rigid2d layer (rigid2d/test/test_diff_drive.cpp,
rigid2d/include/rigid2d/rigid2d.hpp:111-138), at the tolerances those tests use."""
import os

import numpy as np
import pytest

import oracle_api as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gm():
    return np.load(os.path.join(GOLD, "ref_gridmapper.npz"))


@pytest.fixture(scope="module")
def rg():
    return np.load(os.path.join(GOLD, "ref_rigid2d.npz"))


@pytest.mark.parametrize("tag", ["g80", "g120"])
def test_gridmapper_scan_sequence_matches_reference_fixture(gm, tag):
    g = orc.GridAPI("orc", grid=tuple(gm[f"{tag}_grid"]))
    assert [g.xsize, g.ysize] == gm[f"{tag}_size"].tolist()
    assert np.array_equal(g.constants(), gm[f"{tag}_constants"])
    scans, poses = gm[f"{tag}_scans"], gm[f"{tag}_poses"]
    for n, ps in enumerate(gm[f"{tag}_ep_pose"]):
        xy = g.end_points(scans[0], ps)
        assert np.array_equal(xy, gm[f"{tag}_ep{n}_xy"])
        assert [g.world2rowmajor(x, y) for x, y in xy] == gm[f"{tag}_ep{n}_idx"].tolist()
    for s in range(5):
        lik = np.array([g.likelihood(scans[s], c)[0] for c in gm[f"{tag}_s{s}_cand"]])
        assert np.array_equal(lik, gm[f"{tag}_s{s}_lik"])
        assert g.integrate_scan(scans[s], poses[s]) == 0
        if s in (0, 1, 4):
            d = g.dump()
            lo = np.zeros(g.G); lo[gm[f"{tag}_s{s}_lo_idx"]] = gm[f"{tag}_s{s}_lo_val"]
            assert np.array_equal(d["log_odds"], lo)
            assert np.array_equal(d["state"], gm[f"{tag}_s{s}_state"].astype(np.int32))
            assert np.array_equal(d["prob"], gm[f"{tag}_s{s}_prob"])
            assert np.array_equal(d["occ_dist"], gm[f"{tag}_s{s}_occ_dist"])
            assert np.array_equal(g.occ_cells(), gm[f"{tag}_s{s}_occ_cells"])
            assert np.array_equal(g.grid_map(), gm[f"{tag}_s{s}_gridmap"])


def test_bresenham_free_cells_match_reference_fixture(gm):
    g = orc.GridAPI("orc", grid=(0.05, -2.0, 2.0, -2.0, 2.0))
    off = 0
    for pt, n in zip(gm["bres_pts"], gm["bres_len"]):
        cells = g.free_index(pt, gm["bres_pose"])
        assert np.array_equal(cells, gm["bres_cells"][off:off + n]), pt
        off += n
    assert off == gm["bres_cells"].size


def test_rigid2d_matches_reference_fixture(rg):
    r = orc.RigidAPI("orc")
    assert np.array_equal([r.normalize_angle_PI(a) for a in rg["ang_in"]], rg["ang_out"])
    for i, (p, q, v, tw) in enumerate(zip(rg["P"], rg["Q"], rg["V"], rg["TW"])):
        assert np.array_equal(r.compose(p, q), rg["compose"][i])
        assert np.array_equal(r.apply(p, v), rg["apply"][i])
        assert np.array_equal(r.inv(p), rg["inv"][i])
        assert np.array_equal(r.integrate_twist(p, tw), rg["twist"][i])
    d = r.dd_create([0, 0, 0], 0.16, 0.033)
    for cmd, st in zip(rg["dd_cmds"], rg["dd_states"]):
        if cmd[0] == 0.0:
            r.dd_feedforward(d, [cmd[1], cmd[2], 0.0])
        else:
            r.dd_update_odometry(d, cmd[1], cmd[2])
        assert np.array_equal(r.dd_state(d), st)
    r.dd_destroy(d)
    for p, w, want in zip(rg["arc_pose"], rg["arc_wheels"], rg["arc_out"]):   # the arc-rollout plant step (N4)
        got, rc = r.dd_arc_step(0.16, 0.033, 0.01, p, w)
        assert rc == 0 and np.array_equal(got, want)
    assert np.array_equal([r.log_odds_to_prob(l) for l in rg["knife_l"]], rg["knife_prob"])
    assert np.array_equal([r.pdf_normal(a, b)[0] for a, b in rg["pdf_in"]], rg["pdf_out"])


def test_knife_edge_thresholds(rg):
    """One occupied hit lands exactly on 0.9 and one free hit exactly on 0.35 with glibc
    (grid_mapper.cpp:440-460 compares prob >= 0.90 / <= 0.35): the values the device's log-odds
    cut-offs are derived from."""
    l, p = rg["knife_l"], rg["knife_prob"]
    assert p[0] >= 0.9 and p[3] <= 0.35 and p[9] == 0.5
    assert 0.35 < p[6] < 0.9          # occ then free: unknown again
    assert p[7] <= 0.35 and p[8] >= 0.9


# ---- known answers held by the reference's own tests ------------------------------------------------
def test_reference_static_asserts_normalize_angle():  # rigid2d.hpp:126-129
    r = orc.RigidAPI("orc")
    pi = np.pi
    assert abs(r.normalize_angle_PI(3.0 / 2.0 * pi) - (-pi / 2.0)) < 1e-12
    assert abs(r.normalize_angle_PI(7.0 / 6.0 * pi) - (-5.0 / 6.0 * pi)) < 1e-12
    assert abs(r.normalize_angle_PI(8.0 / 3.0 * pi) - (2.0 / 3.0 * pi)) < 1e-12
    assert abs(r.normalize_angle_PI(np.deg2rad(350)) - r.normalize_angle_PI(np.deg2rad(-10))) < 1e-12


def test_reference_gtest_diff_drive_known_answers():  # rigid2d/test/test_diff_drive.cpp:14-386
    r = orc.RigidAPI("orc")
    d = r.dd_create([0, 0, 0], 1.0, 0.02)
    for tw, exp in (([1, 0, 0], (-25, 25)), ([0, 1, 0], (50, 50)), ([1, 1, 0], (25, 75))):
        assert np.allclose(r.dd_twist_to_wheels(d, tw)[0], exp, atol=1e-6)
    r.dd_destroy(d)
    d = r.dd_create([0, 0, 0], 1.0, 0.02)
    v = r.dd_update_odometry(d, np.pi / 30, np.pi / 30)            # PureTranslationOdom
    assert np.allclose(v, 0.10472, atol=1e-3) and np.allclose(r.dd_state(d)[:3], [0, 0.0020944, 0], atol=1e-3)
    r.dd_destroy(d)
    d = r.dd_create([0, 0, 0], 1.0, 0.02)
    v = r.dd_update_odometry(d, -np.pi / 30, np.pi / 30)           # PureRotationOdom
    assert np.allclose(v, [-0.10472, 0.10472], atol=1e-3) and np.allclose(r.dd_state(d)[:3], [0.00418879, 0, 0], atol=1e-3)
    r.dd_destroy(d)
    for tw, exp in (([0, 0.01, 0], [0, 0.01, 0]), ([np.pi / 10, 0, 0], [0.314159, 0, 0]),
                    ([np.pi / 10, 0.01, 0], [0.314159, 0.00983632, 0.00155792])):   # *FeedForward
        d = r.dd_create([0, 0, 0], 1.0, 0.02)
        assert r.dd_feedforward(d, tw) == 0
        assert np.allclose(r.dd_state(d)[:3], exp, atol=1e-3)
        r.dd_destroy(d)


# ---- particle filter restatement: structural checks (parity unpinned: needs Eigen) ----------------
def test_pf_first_scan_clamps_and_draw_count():
    """First scan: maps are empty so likelihoodFieldModel returns 1.0 (grid_mapper.cpp:94-98), both
    likelihood clamps sit at their minimum 1.0, eta = k, the proposal mean is the sample mean."""
    p = orc.pf_params(N=6, k=10)
    pf = orc.PfAPI(p)
    rng = np.random.default_rng(0)
    scan = orc.room_scan((0, 0, 0), walls=(-1.6, 1.5, -1.3, 1.7), rng=rng)
    nz = orc.normal_stream(5, pf.normals_per_scan(True), 0.0, 1.0)
    tr = pf.slam(scan, (0, 0, 0), (0.01, 0.02, 0.0), (0, 0, 0), True, (0.01, 0.02, 0.0), nz)
    assert tr["rc"] == 0 and np.all(tr["p_scan"] == 1.0)
    assert np.all(tr["eta"] == np.clip(tr["p_pose"], 1.0, 10.0).sum(1))
    assert tr["normals_used"] == 6 * 33 + tr["resampled"]
    pose, prev, w = pf.particles()
    assert abs(w.sum() - 1.0) < 1e-12 and np.all(prev == 0.0)
    assert np.allclose(pose[:, 1:], [0.02, 0.0], atol=1e-3)
    assert len(pf.grid(0).occ_cells()) > 100
    pf.close()


@pytest.mark.parametrize("z", [-1.5, 0.0, 0.4, 1.5])
def test_low_variance_resampling_quirks(z):
    """r ~ N(0,1)/N may be negative, comb spacing is 1/(N-1), index clamps at N-1, weights are not
    reset (particle_filter.cpp:468-500)."""
    N = 8
    p = orc.pf_params(N=N, k=4)
    pf = orc.PfAPI(p)
    w = np.array([0.9, 0.02, 0.02, 0.02, 0.01, 0.01, 0.01, 0.01])
    pf.set_particles(w=w)
    scan = np.full(360, 5.0, dtype=np.float32)  # every beam gated out: map untouched, likelihood 1
    nz = np.zeros(pf.normals_per_scan(True)); nz[-1] = z
    # the odometry must move: identical poses give zero variance and the reference throws (pdfNormal)
    tr = pf.slam(scan, (0, 0.05, 0), (0.0, 0.05, 0.0), (0, 0, 0), True, (0.0, 0.05, 0.0), nz)
    assert tr["rc"] == 0 and tr["resampled"] == 1 and tr["neff"] == 1
    idx = tr["resample_idx"]
    assert idx[0] == 0 and np.all(np.diff(idx) >= 0)
    if z == 1.5:
        assert idx[-1] == N - 1   # U_{N-1} = r + 1 > sum of weights: runs off the end and clamps
    if z == -1.5:
        assert np.all(idx == 0)   # negative offset: the comb never leaves particle 0 (w0 = 0.9)
    # hand evaluation of the same loop
    wn = w * tr["eta"]; wn = wn / wn.sum()
    exp, i, c = [], 0, wn[0]
    for m in range(N):
        U = z / N + m * (1.0 / (N - 1))
        while U > c:
            i += 1
            if i > N - 1:
                i = N - 1; break
            c += wn[i]
        exp.append(i)
    assert idx.tolist() == exp
    pf.close()


# ---- frozen vectors of the two Eigen-dependent pieces (tests/golden/make_golden_paths.py:
#      outputs of the restatement, NOT of the reference — they freeze the oracle and give the second restatement and the HIP
#      path something to be held against that does not move) ------------------------------------------------------------------
import zlib  # noqa: E402

import second_restatement as sr  # noqa: E402
from cases import WAYPOINTS, mppi_cfg  # noqa: E402


def _crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


@pytest.fixture(scope="module")
def gp_mppi():
    return np.load(os.path.join(GOLD, "path_mppi.npz"))


@pytest.fixture(scope="module")
def gp_rbpf():
    return np.load(os.path.join(GOLD, "path_rbpf.npz"))


def test_mppi_frozen_vectors_G_A1_G_A2(gp_mppi):
    g = gp_mppi
    d = mppi_cfg(64, 0.25)
    for t in range(3):
        args = (g[f"a1_u_before_{t}"], (0.0, 0.0), tuple(g["a1_xd"]), tuple(g[f"a1_x0_{t}"]), g[f"a1_noise_{t}"])
        r = orc.mppi_new_controls(d, *args)
        for key in ("loss", "J"):                                     # the oracle: bit for bit
            assert np.array_equal(r[key], g[f"a1_{key}_{t}"]), (t, key)
        assert np.array_equal(r["u"], g[f"a1_u_after_{t}"]) and np.array_equal(np.array(r["out"]), g[f"a1_out_{t}"])
        r2 = sr.mppi_new_controls(d, *args)                           # the second restatement: to rounding, without the oracle
        assert np.allclose(r2["loss"], g[f"a1_loss_{t}"], rtol=1e-12, atol=0) and np.allclose(r2["J"], g[f"a1_J_{t}"], rtol=1e-12, atol=0)
        assert np.allclose(r2["u"], g[f"a1_u_after_{t}"], rtol=1e-11, atol=1e-13)
        if t:
            assert np.array_equal(g[f"a1_u_before_{t}"], g[f"a1_u_after_{t - 1}"])  # the warm start really is the shifted vector
    d = mppi_cfg(1024, 0.5)
    noise = orc.normal_stream(int(g["a2_seed"]), 1024 * 50 * 2, 0.0, np.sqrt(d["ul_var"])).reshape(1024, 50, 2)
    assert _crc(noise) == g["a2_noise_crc"]
    r = orc.mppi_new_controls(d, np.zeros((2, 50)), (0.0, 0.0), WAYPOINTS[1], (0.0, 0.0, 0.0), noise)
    assert _crc(r["J"]) == g["a2_J_crc"] and np.array_equal(r["u"], g["a2_u_after"]) and np.array_equal(np.array(r["out"]), g["a2_out"])


def test_particle_filter_frozen_trace_G_B4(gp_rbpf):
    from golden.make_golden_paths import RBPF_SCENARIO as S
    g = gp_rbpf
    N, k = S["N"], S["k"]
    pf = orc.PfAPI(orc.pf_params(N=N, k=k))
    for s in range(S["n_scans"]):
        prev, cur, t_icp, u = g[f"b4_odom_{s}"]
        normals = orc.normal_stream(S["normals_seed"] + s, pf.normals_per_scan(True), 0.0, 1.0)
        assert _crc(normals) == g[f"b4_normals_crc_{s}"]
        if s == S["force_resample_at"]:
            pf.set_particles(w=g["b4_forced_w"])
        tr = pf.slam(g[f"b4_scan_{s}"], u, cur, prev, True, t_icp, normals)
        assert tr["rc"] == 0
        for key in ("sampled", "p_scan", "p_pose", "mu", "sigma", "eta", "new_pose", "weight_raw"):
            assert np.array_equal(tr[key], g[f"b4_{key}_{s}"]), (s, key)
        assert tr["neff"] == g[f"b4_neff_{s}"] and tr["resampled"] == g[f"b4_resampled_{s}"]
        if tr["resampled"]:
            assert np.array_equal(tr["resample_idx"], g[f"b4_parents_{s}"])
        pose, _, w = pf.particles()
        assert np.array_equal(pose, g[f"b4_pose_after_{s}"]) and np.array_equal(w, g[f"b4_weight_after_{s}"])
    assert pf.best() == g["b4_best"] and np.array_equal(pf.grid(pf.best()).dump()["log_odds"], g["b4_log_odds_best"])
    pf.close()


def test_low_variance_resampling_frozen_lists_G_B5(gp_rbpf):
    g = gp_rbpf
    for i in range(5):
        w = g[f"b5_w_{i}"]
        wn, _, _, neff, resample = sr.normalize_and_neff(w.copy())
        assert neff == g[f"b5_neff_{i}"]
        for j, z in enumerate(g["b5_offsets"]):
            want = g[f"b5_idx_{i}_{j}"]
            assert np.array_equal(sr.low_variance_resampling(wn, float(z)), want)
            assert np.all(np.diff(want) >= 0) and want.min() >= 0 and want.max() <= len(w) - 1  # non-decreasing, clamped
            if resample:  # the oracle reaches the selection only when Neff triggers it
                pfr = orc.PfAPI(orc.pf_params(N=len(w), k=4))
                pfr.set_particles(w=w)
                nz = np.zeros(pfr.normals_per_scan(False)); nz[-1] = z
                tr = pfr.slam(np.zeros(360, dtype=np.float32), (0, 0, 0), (0, 0, 0), (0, 0, 0), False, (0, 0, 0), nz)
                assert tr["resampled"] == 1 and np.array_equal(tr["resample_idx"], want)
                pfr.close()
