"""Generates tests/golden/path_{mppi,rbpf}.npz: SURVEY.md 8-c's fixtures G-A1, G-A2, G-B4, G-B5 for the two Eigen-dependent
pieces of the path.

WHAT THESE ARE — and are not.  The reference's mppi.cpp / rk4.cpp / particle_filter.cpp cannot be compiled in this image (Eigen
3.3 is absent and stand-ins are ruled out), so these vectors are NOT outputs of the reference: they are outputs of the C++
restatement (oracle/mppi_oracle.cpp, oracle/rbpf_oracle.cpp — whose GridMapper part IS pinned bit-exact to the compiled
reference), written down once it agreed with the independent second restatement (tests/second_restatement.py; this script
refuses to write a fixture on which the two disagree).  They freeze the oracle: an edit to either restatement that changes
a bit shows up in tests/test_oracle_golden.py, and the `-m gpu` tests hold the HIP path against them without the oracle in the loop.
"parity unpinned" stays true for these pieces.

  python tests/golden/make_golden_paths.py        (run from the repo root, after __graft_entry__.build())
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_api as orc  # noqa: E402
import rbpf_cases as rc  # noqa: E402
import second_restatement as sr  # noqa: E402
from cases import WAYPOINTS, mppi_cfg  # noqa: E402


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def mppi_fixtures():
    out = {}
    # G-A1: cfg1 (K = 64, T = 25), three consecutive calls (warm-start shift), everything stored
    d = mppi_cfg(64, 0.25)
    T = orc.mppi_steps(d)
    u = np.zeros((2, T)); u2 = np.zeros((2, T))
    x0s = [(0.0, 0.0, 0.0), (0.004, 0.0005, 0.02), (0.009, 0.0012, 0.045)]
    for t, x0 in enumerate(x0s):
        noise = orc.normal_stream(42 + t, 64 * T * 2, 0.0, np.sqrt(d["ul_var"])).reshape(64, T, 2)
        r = orc.mppi_new_controls(d, u, (0.0, 0.0), WAYPOINTS[1], x0, noise)
        r2 = sr.mppi_new_controls(d, u2, (0.0, 0.0), WAYPOINTS[1], x0, noise)
        assert np.allclose(r2["J"], r["J"], rtol=1e-12, atol=0) and np.allclose(r2["u"], r["u"], rtol=1e-11, atol=1e-13)
        out.update({f"a1_x0_{t}": np.array(x0), f"a1_u_before_{t}": u.copy(), f"a1_noise_{t}": noise, f"a1_loss_{t}": r["loss"],
                    f"a1_J_{t}": r["J"], f"a1_u_after_{t}": r["u"], f"a1_out_{t}": np.array(r["out"])})
        u, u2 = r["u"], r2["u"]
    out["a1_xd"] = np.array(WAYPOINTS[1])
    # G-A2: one cfg2 call (K = 1024, T = 50): seed only + outputs and a checksum of J
    d = mppi_cfg(1024, 0.5)
    T = orc.mppi_steps(d)
    noise = orc.normal_stream(42, 1024 * T * 2, 0.0, np.sqrt(d["ul_var"])).reshape(1024, T, 2)
    r = orc.mppi_new_controls(d, np.zeros((2, T)), (0.0, 0.0), WAYPOINTS[1], (0.0, 0.0, 0.0), noise)
    r2 = sr.mppi_new_controls(d, np.zeros((2, T)), (0.0, 0.0), WAYPOINTS[1], (0.0, 0.0, 0.0), noise)
    assert np.allclose(r2["J"], r["J"], rtol=1e-12, atol=0) and np.allclose(r2["out"], r["out"], rtol=1e-11, atol=1e-13)
    out.update(a2_seed=np.int64(42), a2_out=np.array(r["out"]), a2_u_after=r["u"], a2_J_crc=crc(r["J"]), a2_J_row0=r["J"][0].copy(),
               a2_J_last=r["J"][-1].copy(), a2_noise_crc=crc(noise))
    return out


RBPF_SCENARIO = dict(N=40, k=50, n_scans=5, inc=(0.04, 0.03, 0.02), scan_seed=3, normals_seed=900, force_resample_at=3)


def rbpf_fixtures():
    """G-B4: the reference's launch configuration (slam.launch:19-42: 40 particles, k = 50, 80 x 80 @ 0.05 m), 5 scans, injected
    ICP, seeded draws, the weights skewed before scan 3 so that resampling fires."""
    S = RBPF_SCENARIO
    N, k = S["N"], S["k"]
    pf = orc.PfAPI(orc.pf_params(N=N, k=k))
    steps, poses = rc.trajectory(S["n_scans"], inc=S["inc"])
    rng = np.random.default_rng(S["scan_seed"])
    out = {}
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        scan = orc.room_scan(poses[s], walls=rc.ROOM_SMALL, rng=rng)
        normals = orc.normal_stream(S["normals_seed"] + s, pf.normals_per_scan(True), 0.0, 1.0)
        if s == S["force_resample_at"]:
            w = np.full(N, 0.2 / N); w[3] += 0.5; w[N // 2] += 0.3; w /= w.sum()
            pf.set_particles(w=w)
            out["b4_forced_w"] = w
        tr = pf.slam(scan, u, cur, prev, True, t_icp, normals)
        assert tr["rc"] == 0
        pose, prev_pose, w_after = pf.particles()
        out.update({f"b4_scan_{s}": scan, f"b4_odom_{s}": np.stack([prev, cur, t_icp, u]), f"b4_normals_crc_{s}": crc(normals),
                    f"b4_sampled_{s}": tr["sampled"], f"b4_p_scan_{s}": tr["p_scan"], f"b4_p_pose_{s}": tr["p_pose"],
                    f"b4_mu_{s}": tr["mu"], f"b4_sigma_{s}": tr["sigma"], f"b4_eta_{s}": tr["eta"], f"b4_new_pose_{s}": tr["new_pose"],
                    f"b4_weight_raw_{s}": tr["weight_raw"], f"b4_neff_{s}": np.int32(tr["neff"]), f"b4_resampled_{s}": np.int32(tr["resampled"]),
                    f"b4_parents_{s}": tr["resample_idx"].copy(), f"b4_pose_after_{s}": pose, f"b4_weight_after_{s}": w_after})
    out["b4_best"] = np.int32(pf.best())
    out["b4_log_odds_best"] = pf.grid(pf.best()).dump()["log_odds"]
    assert sum(int(out[f"b4_resampled_{s}"]) for s in range(S["n_scans"])) >= 1
    pf.close()
    # G-B5: lowVarianceResampling on five weight vectors x offsets (negative r, clamp at the end, a dominant particle)
    rng = np.random.default_rng(5)
    vecs = [np.full(8, 1.0), rng.random(40), np.r_[np.full(39, 1e-3), 5.0][rng.permutation(40)], np.r_[np.full(8, 1e-6), 1.0],
            np.r_[3.0, np.full(30, 0.01)]]
    for i, w in enumerate(vecs):
        wn, _, _, neff, _ = sr.normalize_and_neff(w.copy())
        out[f"b5_w_{i}"] = w
        out[f"b5_neff_{i}"] = np.int32(neff)
        for j, z in enumerate((-2.0, -0.3, 0.0, 0.7, 2.5)):
            N = len(w)
            pfr = orc.PfAPI(orc.pf_params(N=N, k=4))
            pfr.set_particles(w=w)
            nz = np.zeros(pfr.normals_per_scan(False)); nz[-1] = z
            tr = pfr.slam(np.zeros(360, dtype=np.float32), (0, 0, 0), (0, 0, 0), (0, 0, 0), False, (0, 0, 0), nz)
            idx2 = sr.low_variance_resampling(wn, z)
            if tr["resampled"]:
                assert np.array_equal(tr["resample_idx"], idx2)
            assert tr["neff"] == neff
            out[f"b5_idx_{i}_{j}"] = idx2  # the selection for this (weights, offset), whether or not Neff would have triggered it
            pfr.close()
    out["b5_offsets"] = np.array([-2.0, -0.3, 0.0, 0.7, 2.5])
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "path_mppi.npz"), **mppi_fixtures())
    np.savez_compressed(os.path.join(HERE, "path_rbpf.npz"), **rbpf_fixtures())
    for f in ("path_mppi.npz", "path_rbpf.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
