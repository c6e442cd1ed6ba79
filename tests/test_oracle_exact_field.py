"""The oracle's `exact_field` switch and windowed cell store (oracle/rbpf_oracle.cpp: Grid::exact_dist, CellStore).

They exist so that the device's DEFAULT distance-lookup mode — the exact distance to the nearest occupied cell, not the
reference's brushfire — can be held against the restated filter at full size (tests/test_rbpf_field_gpu.py).  Here, on the
CPU, the switch itself is pinned to things that are already pinned:
  * exact_dist == sqrt(exact_edt_codes) * resolution, the brute-force transform the device EDT is tested against, injected
    into the (reference-pinned) GridMapper's own likelihoodFieldModel through occ_dist — bit for bit;
  * a windowed filter == the dense filter, bit for bit, and leaving the window is an error, not silence;
  * with the switch off nothing changed (the golden / reference tests cover that).
"""
import ctypes as C

import numpy as np
import pytest

import oracle_api as orc
import rbpf_cases as rc


def _exact_occ_dist(g: orc.GridAPI) -> np.ndarray:
    occ = np.zeros(g.G, dtype=np.uint8)
    occ[g.occ_cells()] = 1
    never = np.full((g.xsize, g.ysize), 0xFFFF, np.uint16)
    codes = orc.exact_edt_codes(occ.reshape(g.xsize, g.ysize), int(g.constants()[3]), never)
    return np.where(codes == 0xFFFF, 10.0, np.sqrt(codes.astype(np.float64)) * 0.05).ravel()


def test_exact_dist_equals_the_brute_force_transform_injected_into_the_pinned_likelihood():
    g = orc.GridAPI("orc")
    _, poses = rc.trajectory(4, inc=(0.04, 0.03, 0.02))
    rng = np.random.default_rng(1)
    scans = [orc.room_scan(p, walls=rc.ROOM_SMALL, rng=rng) for p in poses]
    probe = np.random.default_rng(2)
    differs = 0
    for s, p in enumerate(poses):
        assert g.integrate_scan(scans[s], p) == 0
        a, b = g.clone(), g.clone()
        a.set_occ_dist(_exact_occ_dist(g))                      # the pinned likelihood over the exact field
        orc.lib().orc_gm_set_exact_field(b.h, C.c_int(1))       # the switch
        for _ in range(25):
            q = np.array(p) + probe.normal(0.0, [0.05, 0.08, 0.08])
            la, ea = a.likelihood(scans[s], q)
            lb, eb = b.likelihood(scans[s], q)
            assert ea == eb == 0 and la == lb, (s, q, la, lb)
            differs += int(g.likelihood(scans[s], q)[0] != lb)
        a.close(); b.close()
    assert differs > 0   # ... and it is NOT the brushfire's likelihood: the switch does something
    g.close()


def _run(pf, n_scans, seed):
    steps, poses = rc.trajectory(n_scans, inc=(0.04, 0.03, 0.02), start=(0.1, 0.033, 0.021))
    rng = np.random.default_rng(seed)
    out = []
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        scan = orc.room_scan(poses[s], walls=rc.ROOM_SMALL, rng=rng)
        normals = orc.normal_stream(70 + s, pf.normals_per_scan(True), 0.0, 1.0)
        if s == 2:
            w = np.full(pf.N, 0.2 / pf.N); w[1] += 0.6; w[pf.N - 2] += 0.2
            pf.set_particles(w=w / w.sum())
        tr = pf.slam(scan, u, cur, prev, True, t_icp, normals)
        out.append(tr)
    return out


def test_windowed_store_is_the_dense_store_and_leaving_the_window_is_an_error():
    N, k = 12, 10
    dense = orc.PfAPI(orc.pf_params(N=N, k=k), exact_field=True)
    win = orc.PfAPI(orc.pf_params(N=N, k=k), exact_field=True, window=(2, 78, 3, 77))
    ta, tb = _run(dense, 4, 5), _run(win, 4, 5)
    assert ta[2]["resampled"] == tb[2]["resampled"] == 1
    for a, b in zip(ta, tb):
        assert a["rc"] == b["rc"] == 0
        for key in ("sampled", "p_scan", "p_pose", "mu", "sigma", "eta", "new_pose", "weight_raw", "resample_idx"):
            assert np.array_equal(a[key], b[key]), key
        assert (a["neff"], a["resampled"]) == (b["neff"], b["resampled"])
    for p in (0, N - 1):
        da, db = dense.grid(p).dump(), win.grid(p).dump()
        assert np.array_equal(da["log_odds"], db["log_odds"]) and np.array_equal(da["state"], db["state"])
    # the room's walls lie outside this window: the first scan writes a cell that has no storage
    small = orc.PfAPI(orc.pf_params(N=2, k=3), exact_field=True, window=(30, 50, 30, 50))
    assert _run(small, 1, 5)[0]["rc"] == 100
    assert not orc.lib().orc_pf_create_ex(C.byref(orc.pf_params(N=2, k=3)), C.c_int(0), orc._p(np.array([0, 8, 0, 8], np.int32)))
    for f in (dense, win, small):
        f.close()


def test_exact_field_filter_differs_from_the_reference_filter_only_through_the_likelihoods():
    """Same draws, same scans: sampled poses and p_pose do not see the field (identical), p_scan does."""
    N, k = 8, 10
    ref, ex = orc.PfAPI(orc.pf_params(N=N, k=k)), orc.PfAPI(orc.pf_params(N=N, k=k), exact_field=True)
    ta, tb = _run(ref, 2, 9), _run(ex, 2, 9)
    assert np.array_equal(ta[0]["p_scan"], tb[0]["p_scan"])          # first scan: empty maps, likelihood 1 on both sides
    assert np.array_equal(ta[1]["sampled"], tb[1]["sampled"]) and np.array_equal(ta[1]["p_pose"], tb[1]["p_pose"])
    assert np.allclose(ta[1]["p_scan"], tb[1]["p_scan"], rtol=0.05)  # near the walls the two fields are close, not equal
    for p in range(N):  # the maps do not depend on the field
        assert np.array_equal(ref.grid(p).dump()["log_odds"], ex.grid(p).dump()["log_odds"])
    ref.close(); ex.close()
