"""Pin the restated oracle (oracle/rbpf_oracle.cpp) against the REAL reference classes compiled from
the reference's own sources (oracle/_ref/libtbnav_ref.so): bit-exact on every output.  Skipped
where the reference build is absent; the committed fixtures in tests/golden/ (generated from the
same library by tests/golden/make_golden.py) carry the same pin everywhere else."""
import numpy as np
import pytest

import oracle_api as orc

pytestmark = pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref not built (needs /root/reference)")


def _pair(**kw):
    return orc.GridAPI("orc", **kw), orc.GridAPI("ref", **kw)


def test_normalize_angle_and_scalar_helpers():
    a, b = orc.RigidAPI("orc"), orc.RigidAPI("ref")
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-50, 50, 2000), [0.0, np.pi, -np.pi, 3 * np.pi, 1e-300, 7.0 / 6.0 * np.pi]])
    for x in xs:
        assert a.normalize_angle_PI(x) == b.normalize_angle_PI(x)
    for l in rng.uniform(-8, 8, 500):
        assert a.log_odds_to_prob(l) == b.log_odds_to_prob(l)
    for p in (0.5, 0.9, 0.35, 0.123):
        assert a.prob_to_log_odds(p) == b.prob_to_log_odds(p)
    for x, v in zip(rng.normal(0, 2, 300), rng.uniform(1e-3, 4, 300)):
        assert a.pdf_normal(x, v) == b.pdf_normal(x, v)
    assert a.pdf_normal(0.3, 1e-13)[1] != 0 and b.pdf_normal(0.3, 1e-13)[1] != 0  # "Variance in pdfNormal is 0"


def test_transform2d_ops_bit_exact():
    a, b = orc.RigidAPI("orc"), orc.RigidAPI("ref")
    rng = np.random.default_rng(1)
    for _ in range(300):
        p, q = rng.uniform(-4, 4, 3), rng.uniform(-4, 4, 3)
        v = rng.uniform(-5, 5, 2)
        tw = rng.uniform(-2, 2, 3)
        assert np.array_equal(a.make(p), b.make(p))
        assert np.array_equal(a.compose(p, q), b.compose(p, q))
        assert np.array_equal(a.apply(p, v), b.apply(p, v))
        assert np.array_equal(a.inv(p), b.inv(p))
        assert np.array_equal(a.integrate_twist(p, tw), b.integrate_twist(p, tw))
    for tw in ([0, 0, 0], [0, 1.5, 0], [0, 0.3, -0.2], [1e-13, 0.2, 0], [0.7, 0, 0]):
        assert np.array_equal(a.integrate_twist([0.3, 1, 2], tw), b.integrate_twist([0.3, 1, 2], tw))


def test_diff_drive_sequences_bit_exact():
    a, b = orc.RigidAPI("orc"), orc.RigidAPI("ref")
    rng = np.random.default_rng(2)
    da, db = a.dd_create([0.1, 0.2, -0.3], 0.16, 0.033), b.dd_create([0.1, 0.2, -0.3], 0.16, 0.033)
    enc = np.zeros(2)
    for step in range(400):
        if step % 3 == 0:
            tw = [rng.uniform(-1, 1), rng.uniform(-0.3, 0.3), 0.0]
            assert a.dd_feedforward(da, tw) == b.dd_feedforward(db, tw) == 0
        else:
            enc = enc + rng.uniform(-0.4, 0.6, 2)
            assert np.array_equal(a.dd_update_odometry(da, *enc), b.dd_update_odometry(db, *enc))
        assert np.array_equal(a.dd_state(da), b.dd_state(db))
        w = rng.uniform(-6, 6, 2)
        assert np.array_equal(a.dd_wheels_to_twist(da, w), b.dd_wheels_to_twist(db, w))
        ta, ra = a.dd_twist_to_wheels(da, [w[0], w[1], 0.0]); tb, rb = b.dd_twist_to_wheels(db, [w[0], w[1], 0.0])
        assert np.array_equal(ta, tb) and ra == rb == 0
    assert a.dd_twist_to_wheels(da, [0, 1, 0.5])[1] != 0 and b.dd_twist_to_wheels(db, [0, 1, 0.5])[1] != 0
    a.dd_destroy(da); b.dd_destroy(db)


def test_arc_plant_step_bit_exact():
    """the exact-arc rollout dynamics are the plant's own step, wheelsToTwist(u) * dt ->
    DiffDrive::feedforward.  The restated step (what the arc-dynamics MPPI oracle integrates with) must equal the
    reference's own class bit for bit: turning both ways, straight lines (|w dt| < 1e-12: the screw's translation
    branch), standstill, headings that wrap at +-pi, and chains of steps fed back into themselves."""
    a, b = orc.RigidAPI("orc"), orc.RigidAPI("ref")
    rng = np.random.default_rng(4)
    wb, wr, dt = 0.16, 0.033, 0.01
    cases = [((0.0, 0.0, 0.0), (1.0, 1.0)), ((0.3, -0.2, 3.14159), (6.3, -6.3)), ((1.0, 2.0, -3.1415926), (-2.0, 5.0)),
             ((0.0, 0.0, 1.0), (0.0, 0.0)), ((0.5, 0.5, 0.7), (2.0, 2.0 + 1e-11)), ((0.5, 0.5, 0.7), (1e-13, -1e-13))]
    cases += [(tuple(rng.uniform(-3, 3, 2)) + (rng.uniform(-7, 7),), tuple(rng.uniform(-6.35, 6.35, 2))) for _ in range(400)]
    for pose, wheels in cases:
        pa, ra = a.dd_arc_step(wb, wr, dt, pose, wheels)
        pb, rb = b.dd_arc_step(wb, wr, dt, pose, wheels)
        assert ra == rb == 0 and np.array_equal(pa, pb), (pose, wheels)
    pa = pb = np.array([0.1, -0.4, 3.0])
    for i in range(500):                       # a lap that crosses the +-pi cut several times
        wheels = (4.0 + np.sin(0.05 * i), 6.0)
        pa, _ = a.dd_arc_step(wb, wr, 0.05, pa, wheels)
        pb, _ = b.dd_arc_step(wb, wr, 0.05, pb, wheels)
        assert np.array_equal(pa, pb)
    assert abs(pa[2]) <= np.pi


@pytest.mark.parametrize("grid,delta", [((0.05, -2.0, 2.0, -2.0, 2.0), 1.0), ((0.05, -3.0, 3.0, -3.0, 3.0), 1.0),
                                        ((0.05, -10.0, 10.0, -10.0, 10.0), 1.0), ((0.1, -5.0, 5.0, -5.0, 5.0), 1.0 / 3.0)])
def test_grid_mapper_scan_sequences_bit_exact(grid, delta):
    """integrateScan (raycast, log-odds, states, occupied set in hash order, brushfire distances),
    likelihoodFieldModel, gridMap, laserEndPoints — 8 noisy scans along a trajectory."""
    la = orc.lds01_laser(delta)
    go, gr = _pair(grid=grid, laser=la, trs=(0.05, 0.02, -0.01))
    assert (go.xsize, go.ysize) == (gr.xsize, gr.ysize) and np.array_equal(go.constants(), gr.constants())
    half = grid[2]
    walls = (-0.8 * half, 0.8 * half, -0.7 * half, 0.7 * half) if half <= 3 else (-3.0, 3.0, -2.5, 2.5)
    rng = np.random.default_rng(7)
    pose = np.zeros(3)
    n = int(round(360 / delta))
    for s in range(8):
        scan = orc.room_scan(pose, n_beams=n, beam_delta_deg=delta, walls=walls, rng=rng)
        cand = pose + rng.normal(0, 0.02, 3)
        assert go.likelihood(scan, cand) == gr.likelihood(scan, cand)
        assert np.array_equal(go.end_points(scan, cand), gr.end_points(scan, cand))
        assert go.integrate_scan(scan, pose) == gr.integrate_scan(scan, pose) == 0
        da, db = go.dump(), gr.dump()
        for key in da:
            assert np.array_equal(da[key], db[key]), (s, key)
        assert np.array_equal(go.occ_cells(), gr.occ_cells())
        assert np.array_equal(go.grid_map(), gr.grid_map())
        pose = pose + np.array([0.07, 0.10, 0.05]) * (half / 10.0 if half > 3 else 0.3)
    # copies behave like the original (particle_filter.cpp:495 deep copies)
    co, cr = go.clone(), gr.clone()
    scan = orc.room_scan(pose, n_beams=n, beam_delta_deg=delta, walls=walls, rng=rng)
    assert co.integrate_scan(scan, pose) == cr.integrate_scan(scan, pose) == 0
    assert np.array_equal(co.dump()["occ_dist"], cr.dump()["occ_dist"])
    assert np.array_equal(co.occ_cells(), cr.occ_cells())


def test_bresenham_all_octants_and_degenerate_rays():
    go, gr = _pair(grid=(0.05, -2.0, 2.0, -2.0, 2.0))
    pose = (0.3, 0.013, -0.021)
    rng = np.random.default_rng(3)
    pts = [(0.013 + r * np.cos(t), -0.021 + r * np.sin(t)) for t in np.linspace(0, 2 * np.pi, 97) for r in (0.0, 0.04, 0.5, 1.7)]
    pts += [(0.013, 1.0), (0.013, -1.0), (1.0, -0.021), (-1.0, -0.021), (1.013, 0.979), (-0.987, -1.021), (1.013, -1.021)]
    pts += [tuple(p) for p in rng.uniform(-1.99, 1.99, (300, 2))]
    for pt in pts:
        a, b = go.free_index(pt, pose), gr.free_index(pt, pose)
        assert a is not None and np.array_equal(a, b), pt
    for _ in range(300):
        x0, y0, x1, y1 = rng.integers(0, 80, 4)
        for which in (0, 1, 2):
            if which == 2:
                d = int(rng.integers(-30, 30)); x1, y1 = x0 + d, y0 + (d if rng.random() < 0.5 else -d)
            assert np.array_equal(go.line_cells(which, x0, y0, x1, y1), gr.line_cells(which, x0, y0, x1, y1))
    # outside the world: both throw
    assert go.free_index((2.5, 0.0), pose) is None and gr.free_index((2.5, 0.0), pose) is None
    assert go.world2rowmajor(0, 2.01) == gr.world2rowmajor(0, 2.01) == -1
    for x, y in [(-2.0, -2.0), (2.0, 2.0), (0.0, 0.0), (1.99999, -0.00001), (0.05, 0.1), (0.15, 0.15000000000000002)]:
        assert go.world2rowmajor(x, y) == gr.world2rowmajor(x, y)


def test_out_of_world_scan_reports_the_same_error():
    go, gr = _pair(grid=(0.05, -2.0, 2.0, -2.0, 2.0))
    scan = np.full(360, 3.0, dtype=np.float32)  # end points 3 m away on a +-2 m map
    assert go.integrate_scan(scan, (0, 0, 0)) != 0 and gr.integrate_scan(scan, (0, 0, 0)) != 0
    # likelihood on an EMPTY map returns 1.0 before any index is formed (grid_mapper.cpp:94-98)
    g2o, g2r = _pair(grid=(0.05, -2.0, 2.0, -2.0, 2.0))
    assert g2o.likelihood(scan, (0, 0, 0)) == g2r.likelihood(scan, (0, 0, 0)) == (1.0, 0)
