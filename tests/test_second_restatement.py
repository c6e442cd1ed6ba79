"""N-version check of the two restatements that cannot be pinned to the compiled reference (no Eigen in the image):
oracle/mppi_oracle.cpp and the ParticleFilter logic of oracle/rbpf_oracle.cpp against tests/second_restatement.py, a second
restatement written from the reference's sources in numpy / plain Python.  CPU only.  The scan likelihoods (GridMapper,
pinned bit-exact to the compiled reference) are inputs of the second restatement, not restated again."""
import numpy as np
import pytest

import oracle_api as orc
import second_restatement as sr
from cases import MPPI_BASE, WAYPOINTS, mppi_cfg, rel_err


@pytest.mark.parametrize("K,horizon,ticks", [(64, 0.25, 3), (1024, 0.5, 2), (37, 0.29, 2)])
def test_mppi_tick_two_restatements_agree(K, horizon, ticks):
    """cfg1 (three ticks: warm-start shift), cfg2, and the truncating horizon 0.29/0.01 -> 28 with a ragged K."""
    d = mppi_cfg(K, horizon)
    T = orc.mppi_steps(d)
    assert T == sr.mppi_steps(horizon, d["dt"])
    u_a = np.zeros((2, T)); u_b = np.zeros((2, T))
    x0 = (0.02, -0.01, 0.1)
    for t in range(ticks):
        noise = orc.normal_stream(42 + t, K * T * 2, 0.0, np.sqrt(d["ul_var"])).reshape(K, T, 2)
        a = orc.mppi_new_controls(d, u_a, (0.3, -0.2), WAYPOINTS[1], x0, noise)
        b = sr.mppi_new_controls(d, u_b, (0.3, -0.2), WAYPOINTS[1], x0, noise)
        assert rel_err(b["loss"], a["loss"]) < 1e-12
        assert rel_err(b["J"], a["J"]) < 1e-12
        # the soft-min sums run over K terms in a different order (numpy pairwise vs sequential): 1e-11 on the controls
        assert np.allclose(b["u"], a["u"], rtol=1e-11, atol=1e-13)
        assert np.allclose(b["out"], a["out"], rtol=1e-11, atol=1e-13)
        u_a, u_b = a["u"], b["u"]
        x0 = (x0[0] + 0.01, x0[1] + 0.004, x0[2] + 0.02)


def test_mppi_clamp_and_single_rollout():
    """One rollout: the soft-min weight is 1 whatever the cost, so u += du and the clamp is the only non-linearity."""
    d = mppi_cfg(1, 0.1, max_wheel_vel=0.5)
    T = orc.mppi_steps(d)
    noise = orc.normal_stream(3, T * 2, 0.0, 2.0).reshape(1, T, 2)
    a = orc.mppi_new_controls(d, np.zeros((2, T)), (0.0, 0.0), WAYPOINTS[2], (0, 0, 0), noise)
    b = sr.mppi_new_controls(d, np.zeros((2, T)), (0.0, 0.0), WAYPOINTS[2], (0, 0, 0), noise)
    assert np.allclose(b["u"], a["u"], rtol=0, atol=1e-15) and np.abs(b["u"]).max() == 0.5
    assert rel_err(b["J"], a["J"]) < 1e-12


def _second_pf_scan(p, state, tr, normals, icp_ok, u, cur_od, prev_od, t_icp):
    """One SLAM() of the second restatement on (pose, prev, weight), particle_filter.cpp:141-251.  tr supplies ONLY the
    raw scan likelihoods of the oracle's pinned GridMapper."""
    N, k = p.num_particles, p.k
    pose, prev, w = state
    stride = 3 * k + 3 if icp_ok else 3
    a = (p.srr, p.srt, p.str_, p.stt)
    clamps = (p.scan_min, p.scan_max, p.pose_min, p.pose_max)
    out = dict(sampled=np.zeros((N, k, 3)), p_pose=np.zeros((N, k)), mu=np.zeros((N, 3)), sigma=np.zeros((N, 3, 3)),
               eta=np.zeros(N), new_pose=np.zeros((N, 3)))
    for i in range(N):
        z = normals[i * stride:(i + 1) * stride]
        if not icp_ok:
            prev[i] = pose[i]
            pose[i] = sr.sample_motion_model(u, pose[i], list(p.motion_noise), z)
            w[i] *= tr["p_scan"][i, 0]
            continue
        center = sr.compose(pose[i], t_icp)
        smp = sr.sample_mode(center, list(p.sample_range), z[:3 * k].reshape(k, 3))
        # prev_pose is updated only AFTER the proposal: it is the pose from two scans ago (reference quirk)
        pp = np.array([sr.pose_likelihood_odom(a, smp[j], prev[i], cur_od, prev_od) for j in range(k)])
        mu, sigma, eta = sr.gaussian_proposal(smp, tr["p_scan"][i], pp, clamps)
        new = mu + sr.cholesky_lower(sigma) @ z[3 * k:3 * k + 3]   # theta NOT wrapped (:214)
        prev[i] = pose[i]
        pose[i] = new
        w[i] *= eta
        out["sampled"][i], out["p_pose"][i], out["mu"][i], out["sigma"][i] = smp, pp, mu, sigma
        out["eta"][i], out["new_pose"][i] = eta, new
    out["weight_raw"] = w.copy()
    wn, s, sq, neff, resample = sr.normalize_and_neff(w)
    w[:] = wn
    out.update(sum_w=s, sq_sum=sq, neff=neff, resampled=int(resample), idx=None)
    if resample:
        idx = sr.low_variance_resampling(w, normals[N * stride])
        pose[:], prev[:], w[:] = pose[idx], prev[idx], w[idx]   # deep copies; weights NOT reset (:495)
        out["idx"] = idx
    return out


@pytest.mark.parametrize("icp_pattern", ["ok", "fail", "mixed"])
def test_particle_filter_logic_two_restatements_agree(icp_pattern):
    """Six scans, N = 12, k = 10 on the reference's 80 x 80 launch grid; weights skewed before scan 3 so that resampling
    fires (the parent list must then agree exactly).  Everything Eigen-dependent is compared; GridMapper is shared."""
    N, k = 12, 10
    p = orc.pf_params(N=N, k=k)
    pf = orc.PfAPI(p)
    rng = np.random.default_rng(11)
    state = (np.zeros((N, 3)), np.zeros((N, 3)), np.full(N, 1.0 / N))
    odom = np.zeros(3)
    n_resampled = 0
    for s in range(6):
        icp_ok = {"ok": True, "fail": False, "mixed": s % 2 == 0}[icp_pattern]
        inc = np.array([0.03, 0.05, 0.02])  # (dtheta, dx, dy) of the odometry
        cur = odom + inc
        t_icp = (inc[0], inc[1], inc[2])
        u = (0.03, 0.05, 0.0)
        true_pose = (cur[1], cur[2], cur[0])
        scan = orc.room_scan(true_pose, walls=(-1.6, 1.5, -1.3, 1.7), rng=rng)
        if s == 3:
            wsk = np.full(N, 0.02 / N); wsk[4] += 0.6; wsk[9] += 0.38
            pf.set_particles(w=wsk); state[2][:] = wsk
        nz = orc.normal_stream(100 + s, pf.normals_per_scan(icp_ok), 0.0, 1.0)
        tr = pf.slam(scan, u, cur, odom, icp_ok, t_icp, nz)
        assert tr["rc"] == 0
        mine = _second_pf_scan(p, state, tr, nz, icp_ok, u, cur, odom, t_icp)
        if icp_ok:
            assert np.allclose(mine["sampled"], tr["sampled"], rtol=0, atol=1e-14)
            assert rel_err(mine["p_pose"], tr["p_pose"]) < 1e-9   # glibc atan2 / exp on both sides, different association
            assert rel_err(mine["eta"], tr["eta"]) < 1e-10
            assert np.allclose(mine["mu"], tr["mu"], rtol=0, atol=1e-13)
            assert np.allclose(mine["sigma"], tr["sigma"], rtol=1e-8, atol=1e-22)
            assert np.allclose(mine["new_pose"], tr["new_pose"], rtol=0, atol=1e-12)
        assert rel_err(mine["weight_raw"], tr["weight_raw"]) < 1e-10
        assert mine["neff"] == tr["neff"] and mine["resampled"] == tr["resampled"]
        assert abs(mine["sum_w"] - tr["sum_w"]) <= 1e-10 * abs(tr["sum_w"])
        if mine["resampled"]:
            n_resampled += 1
            assert np.array_equal(mine["idx"], tr["resample_idx"])
        o_pose, o_prev, o_w = pf.particles()
        assert np.allclose(state[0], o_pose, rtol=0, atol=1e-12) and np.allclose(state[1], o_prev, rtol=0, atol=1e-12)
        assert rel_err(state[2], o_w) < 1e-10
        # carry the oracle's state forward bit for bit so that a 1e-13 difference cannot move a later grid index
        state[0][:], state[1][:], state[2][:] = o_pose, o_prev, o_w
        odom = cur
    assert n_resampled >= 1
    pf.close()


@pytest.mark.parametrize("z", [-2.0, -0.3, 0.0, 0.7, 2.5])
def test_low_variance_resampling_two_restatements_agree(z):
    """SURVEY G-B5's cases: negative offset, clamp at the end, a dominant particle."""
    rng = np.random.default_rng(5)
    for N, shape in [(8, "uniform"), (40, "random"), (40, "one_heavy"), (9, "tail_heavy")]:
        w = {"uniform": np.full(N, 1.0), "random": rng.random(N), "one_heavy": np.r_[np.full(N - 1, 1e-3), 5.0][rng.permutation(N)],
             "tail_heavy": np.r_[np.full(N - 1, 1e-6), 1.0]}[shape]
        p = orc.pf_params(N=N, k=4)
        pf = orc.PfAPI(p)
        pf.set_particles(w=w)
        nz = np.zeros(pf.normals_per_scan(False)); nz[-1] = z
        scan = np.zeros(360, dtype=np.float32)  # no valid beam: the likelihood is 1, the weights pass through
        tr = pf.slam(scan, (0, 0, 0), (0, 0, 0), (0, 0, 0), False, (0, 0, 0), nz)
        wn, _, _, neff, resample = sr.normalize_and_neff(w.copy())
        assert neff == tr["neff"] and int(resample) == tr["resampled"]
        if resample:
            assert np.array_equal(sr.low_variance_resampling(wn, z), tr["resample_idx"])
        pf.close()
