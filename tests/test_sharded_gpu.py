"""The sharded drivers with the HIP backends: two ranks (two processes) sharing ONE MI355X, gloo for
the exchange (RCCL needs one GPU per rank; the 8-GPU run is the driver's).  The sharded result must
equal the unsharded HIP run — bit-exact for the RBPF (same kernels per particle, same sequential
normalise/selection), within fp64 reassociation for the MPPI soft-min."""
import numpy as np
import pytest

import oracle_api as orc
from cases import WAYPOINTS, make_mppi, mppi_cfg
from dist_workers import mppi_hip_worker, rbpf_hip_worker, rbpf_scenario, run_spawn

pytestmark = pytest.mark.gpu


def test_mppi_two_ranks_one_gpu(gpu_pkg):
    K, horizon, n_ticks, seed = 4096, 0.5, 3, 9
    res = run_spawn(mppi_hip_worker, 2, K, horizon, n_ticks, seed)
    d = mppi_cfg(K, horizon)
    T = orc.mppi_steps(d)
    m = make_mppi(gpu_pkg, d)
    m.setWaypoint(*WAYPOINTS[1])
    u = np.zeros((2, T))
    for t in range(n_ticks):
        noise = orc.normal_stream(seed + t, K * T * 2, 0.0, np.sqrt(0.9)).reshape(K, T, 2)
        got = m.newControls(0.0, 0.0, 0.0, noise)
        ref = orc.mppi_new_controls(d, u, (0, 0), WAYPOINTS[1], (0, 0, 0), noise)
        u = ref["u"]
        for r in (0, 1):
            assert np.allclose(res[r]["outs"][t], got, rtol=1e-10, atol=1e-13)
            assert np.allclose(res[r]["outs"][t], ref["out"], rtol=1e-9, atol=1e-12)
    assert np.array_equal(res[0]["u"], res[1]["u"])
    assert np.allclose(res[0]["u"], m.getControls(), rtol=1e-10, atol=1e-13)


def _rbpf_sharded_vs_unsharded(n_local, k, skew_scan, heavy, backend, world=2):
    from rtn_amd.rbpf import ParticleFilter, default_params
    N = world * n_local
    res = run_spawn(rbpf_hip_worker, world, n_local, k, skew_scan, heavy, backend=backend)
    pf = ParticleFilter(default_params(N=N, k=k))
    steps, scans = rbpf_scenario(4)
    stride = 3 * k + 3
    resampled_any = False
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        normals = orc.normal_stream(50 + s, N * stride + 1, 0.0, 1.0)
        if s == skew_scan:
            w = np.full(N, 0.01)
            for i, v in heavy.items():
                w[i] = v
            w /= w.sum()
            pf.setParticles(w=w)
        st = pf.SLAM(scans[s], u, cur, prev, True, t_icp, normals)
        for r in range(world):
            neff, resampled, parents = res[r]["hist"][s]
            assert (neff, resampled) == (st.neff, st.resampled)
            if st.resampled:
                assert parents == pf.trace()["resample_idx"].tolist()
        resampled_any |= bool(st.resampled)
    assert resampled_any
    assert sum(res[r]["migrated"] for r in range(world)) > 0  # a particle really crossed ranks ...
    pose, prev_pose, w = pf.particles()
    for r in range(world):
        sl = slice(r * n_local, (r + 1) * n_local)
        assert np.array_equal(res[r]["pose"], pose[sl]) and np.array_equal(res[r]["prev"], prev_pose[sl])
        assert np.array_equal(res[r]["w"], w[sl])
        for p in range(n_local):                            # ... and the scans after the resample kept every map equal
            assert np.array_equal(res[r]["lo"][p], pf.logOdds(r * n_local + p))
            assert np.array_equal(res[r]["codes"][p], pf.distCode(r * n_local + p))
    return res


@pytest.mark.parametrize("heavy", [{3: 0.6, 10: 0.25}, {2: 0.35, 5: 0.55}])
def test_rbpf_two_ranks_one_gpu_equals_unsharded_bit_exact(gpu_pkg, heavy):
    """Two ranks on one GPU (gloo carries the exchange, the HIP handles do the work), a forced cross-rank resample
    after the second scan and two more scans.  {2: .35, 5: .55} with 8 particles per rank is the case where an
    EXPORTED parent (5, sent to rank 1) has its own slot taken over by another local parent (2)."""
    n_local = 6 if 10 in heavy else 8
    res = _rbpf_sharded_vs_unsharded(n_local, 8, 1, heavy, "gloo")
    # a travelling particle is its tiles, not its map: 80 x 80 cells = 9 tiles of 8 KB + 128 B at most + counts + state
    assert max(res[0]["migrated"], res[1]["migrated"]) < 8 * (9 * 8192 + 4096)


def test_rbpf_four_ranks_one_gpu_children_span_three_ranks(gpu_pkg):
    """Four ranks of 5 particles: particle 1 (rank 0) takes ~60 % of the global weight, so its children fill rank 0, rank 1
    and part of rank 2 — one blob goes to two destinations — and particle 17 (rank 3) feeds ranks 2 and 3: a rank that
    both sends and receives, ranks whose every slot is imported, parents shifted across more than one rank boundary."""
    res = _rbpf_sharded_vs_unsharded(5, 8, 1, {1: 0.6, 17: 0.3}, "gloo", world=4)
    parents = np.array(next(h[2] for h in res[0]["hist"] if h[1]))
    assert len({m // 5 for m in np.nonzero(parents == 1)[0]}) >= 3   # particle 1's children live on three ranks
    assert res[0]["migrated"] > 0


@pytest.mark.parametrize("world,n_local,heavy", [(3, 7, {0: 0.3, 9: 0.3, 20: 0.3}), (3, 4, {11: 0.9}), (4, 3, {5: 0.45, 6: 0.45})])
def test_rbpf_more_rank_layouts_equal_unsharded_bit_exact(gpu_pkg, world, n_local, heavy):
    """Other shapes of the exchange: three heavy particles one per rank (every rank sends and receives), one particle of the
    LAST rank taking nearly everything (all ranks import from it, it keeps itself), two neighbours on one rank feeding four."""
    _rbpf_sharded_vs_unsharded(n_local, 8, 1, heavy, "gloo", world=world)


def test_rbpf_two_ranks_two_gpus_rccl(gpu_pkg):
    """The same exchange over RCCL (backend "nccl"), one GPU per rank: device tensors end to end."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the 1-GPU box runs the gloo variant above)")
    _rbpf_sharded_vs_unsharded(8, 8, 1, {2: 0.35, 5: 0.55}, "nccl")
