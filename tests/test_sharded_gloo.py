"""world_size-2 tests of the exchange step of both paths on CPU (gloo, 127.0.0.1).

MPPI: each rank computes its slice's partial records with the oracle, ONE all-gather, every rank
combines — the result equals the unsharded oracle tick (and is identical on both ranks).
RBPF: weights all-gather -> the product's host-side tbnav_rbpf_resample_global (bit-exact against the
oracle filter's normalise/Neff/low-variance selection) -> point-to-point particle migration; every
slot ends up with its parent's pose/weight/maps."""
import numpy as np
import pytest

import oracle_api as orc
from cases import WAYPOINTS, mppi_cfg
from dist_workers import mppi_worker, rbpf_worker, run_spawn


def test_mppi_two_shards_one_all_gather_equals_unsharded():
    K, horizon, n_ticks, seed = 96, 0.25, 3, 17
    res = run_spawn(mppi_worker, 2, K, horizon, n_ticks, seed)
    d = mppi_cfg(K, horizon)
    T = orc.mppi_steps(d)
    u = np.zeros((2, T))
    for t in range(n_ticks):
        noise = orc.normal_stream(seed + t, K * T * 2, 0.0, np.sqrt(0.9)).reshape(K, T, 2)
        ref = orc.mppi_new_controls(d, u, (0, 0), WAYPOINTS[1], (0, 0, 0), noise)
        u = ref["u"]
        for r in (0, 1):
            assert np.allclose(res[r]["outs"][t], ref["out"], rtol=1e-10, atol=1e-13)
    assert np.array_equal(res[0]["u"], res[1]["u"])          # every rank holds the same warm start
    assert np.allclose(res[0]["u"], u, rtol=1e-10, atol=1e-13)


def test_resample_global_is_bit_exact_against_the_oracle_filter(pkg):
    """The exchange step's host routine vs the oracle's normalizeWeights / effectiveParticles /
    lowVarianceResampling (particle_filter.cpp:442-500), driven through the oracle filter."""
    from rtn_amd.rbpf import resample_global
    N, k = 24, 3
    rng = np.random.default_rng(5)
    for trial in range(6):
        pf = orc.PfAPI(orc.pf_params(N=N, k=k))
        w = rng.random(N) ** (1 + 3 * trial)
        w /= w.sum()
        pf.set_particles(w=w)
        scan = np.full(360, 9.0, dtype=np.float32)  # all beams gated: likelihood 1, maps untouched
        nz = rng.standard_normal(pf.normals_per_scan(True))
        tr = pf.slam(scan, (0, 0.05, 0), (0.0, 0.05, 0.0), (0, 0, 0), True, (0.0, 0.05, 0.0), nz)
        assert tr["rc"] == 0
        parents, wn, st = resample_global(tr["weight_raw"], nz[-1])
        assert (st.neff, st.resampled) == (tr["neff"], tr["resampled"])
        assert st.sum_w == tr["sum_w"] and st.sq_sum == tr["sq_sum"]
        _, _, w_after = pf.particles()
        if st.resampled:
            assert np.array_equal(parents, tr["resample_idx"])
            assert np.array_equal(wn[parents], w_after)      # weights follow their parents, not reset
        else:
            assert np.array_equal(parents, np.arange(N)) and np.array_equal(wn, w_after)
        pf.close()


@pytest.mark.parametrize("seed,world", [(1, 2), (2, 2), (3, 2), (4, 3), (5, 4)])
def test_rbpf_shards_weights_all_gather_and_migration(seed, world):
    """world 3 / 4: a rank sends to several destinations (one message each) and receives from several sources into one buffer."""
    n_local = 8 if world == 2 else 5
    res = run_spawn(rbpf_worker, world, n_local, seed)
    N = world * n_local
    for r in range(1, world):
        assert np.array_equal(res[0]["parents"], res[r]["parents"]) and res[0]["neff"] == res[r]["neff"]
    parents = res[0]["parents"]
    # unsharded expectation: the initial global state gathered by `parents`
    gid = np.arange(N)
    rng = np.random.default_rng(seed)
    normals = rng.standard_normal(N + 1)
    raw = np.abs(normals[:N]) ** 4 + 1e-3
    wn = raw / np.sum(raw)  # (summation order aside; compared loosely)
    maps0 = gid[:, None] * 1000.0 + np.arange(32)[None, :]
    state0 = np.stack([gid + 0.25, gid * 2.0, gid * 3.0, gid + 0.5, gid * 5.0, gid * 7.0], 1)
    got_state = np.concatenate([res[r]["state"] for r in range(world)])
    got_maps = np.concatenate([res[r]["maps"] for r in range(world)])
    got_dist = np.concatenate([res[r]["dist"] for r in range(world)])
    assert np.array_equal(got_state[:, :6], state0[parents])
    assert np.array_equal(got_maps, maps0[parents]) and np.array_equal(got_dist, -maps0[parents])
    assert np.allclose(got_state[:, 6], wn[parents], rtol=1e-12)
    if res[0]["resampled"]:
        assert len(set(parents.tolist())) < N
