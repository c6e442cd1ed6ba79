"""GPU tests of the in-library sharded paths (include/tbnav_comm.h, tbnav_mppi_attach_comm / tbnav_mppi_group_*,
tbnav_rbpf_attach_comm / tbnav_rbpf_group_*): the exchange is issued by libtbnav_hip.so itself on the handle's stream —
RCCL between distinct devices, event-ordered copies between members of a one-process group that share a device.  On this
one-GPU box: a REAL RCCL communicator of one rank (ncclCommInitRank + ncclAllGather execute), and groups of 2 ... 8 members
on device 0 for everything the number of ranks changes (record layout, noise counter space, combine over G shards,
cross-member migration)."""
import numpy as np
import pytest

import oracle_api as orc
from cases import WAYPOINTS, make_mppi, mppi_cfg, rel_err

pytestmark = pytest.mark.gpu


def _group(d, devices, **kw):
    from rtn_amd.mppi import MPPIGroup, CartModel, LossFunc
    return MPPIGroup(CartModel(d["wheel_radius"], d["wheel_base"]), LossFunc(d["Q"], d["R"], d["P1"]), d["lam"], d["max_wheel_vel"],
                     d["ul_var"], d["ur_var"], d["horizon"], d["dt"], d["rollouts"], devices=devices, **kw)


def test_rccl_communicator_of_one_rank_carries_the_mppi_tick(gpu_pkg):
    """ncclGetUniqueId -> ncclCommInitRank(nranks = 1) -> a handle with the communicator attached: its ticks run shard
    partials -> ncclAllGather -> combine inside the library.  Against the oracle (host noise) and, bit for bit, against the
    same three steps driven by hand on a second handle (tbnav_mppi_shard_partials / _shard_combine)."""
    import torch
    from rtn_amd.comm import Comm
    comm = Comm.create(Comm.unique_id(), 1, 0, 0)
    assert (comm.rank, comm.size, comm.device, comm.uses_rccl) == (0, 1, 0, True)
    comm.selftest(1 << 20)   # ncclAllGather + ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd (a message to itself) really execute
    for K, horizon in ((1024, 0.5), (40000, 0.24)):
        d = mppi_cfg(K, horizon)
        T = orc.mppi_steps(d)
        a, b = make_mppi(gpu_pkg, d), make_mppi(gpu_pkg, d)
        a.attachComm(comm)
        for m in (a, b):
            m.setWaypoint(*WAYPOINTS[1])
        noise = orc.normal_stream(21, K * T * 2, 0.0, np.sqrt(0.9)).reshape(K, T, 2)
        ref = orc.mppi_new_controls(d, np.zeros((2, T)), (0, 0), WAYPOINTS[1], (0.0, 0.0, 0.1), noise)
        got = a.newControls(0.0, 0.0, 0.1, noise)
        assert np.allclose(got, ref["out"], rtol=1e-9, atol=1e-12) and np.allclose(a.getControls(), ref["u"], rtol=1e-9, atol=1e-12)
        assert rel_err(a.costToGo(), ref["J"]) < 1e-12
        # by hand on b: same kernels, same records
        dn = torch.from_numpy(noise).cuda()
        dl, dr = dn[:, :, 0].t().contiguous(), dn[:, :, 1].t().contiguous()
        rec = torch.zeros(T, b.records_per_step, 8, dtype=torch.float64, device="cuda")
        b.shardPartials((0.0, 0.0, 0.1), dl.data_ptr(), dr.data_ptr(), rec.data_ptr())
        b.shardCombine(rec.data_ptr(), 1)
        torch.cuda.synchronize()
        assert np.array_equal(a.getControls(), b.getControls())
        # production ticks (device noise), a batch through the C loop: the attached handle == by hand
        st = torch.cuda.Stream()
        a.enqueueRngBatch((0.0, 0.0, 0.1), 5, 100, 7, st.cuda_stream)
        for i in range(7):
            b.shardPartialsRng((0.0, 0.0, 0.1), 5, 100 + i, rec.data_ptr(), st.cuda_stream)
            b.shardCombine(rec.data_ptr(), 1, st.cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(a.getControls(), b.getControls())
        a.attachComm(None)   # detached: the plain tick again
        assert np.allclose(a.newControls(0.0, 0.0, 0.1, noise), b.newControls(0.0, 0.0, 0.1, noise), rtol=1e-9, atol=1e-12)
        a.close(); b.close()
    comm.close()


@pytest.mark.parametrize("K,horizon,P", [(1024, 0.5, 2), (4096, 0.25, 4), (8 * 8192, 0.32, 8), (3 * 2048, 1.0, 3)])
def test_mppi_group_of_members_on_one_device_equals_the_oracle_and_the_unsharded_noise(gpu_pkg, K, horizon, P):
    """tbnav_mppi_group with P members on device 0 (copy transport): host-noise ticks against the oracle over the whole
    ensemble (two ticks, warm start), then production ticks: the members draw the ENSEMBLE's perturbations (disjoint slices
    of one counter space), so the group's controls equal the unsharded handle's device-noise tick to rounding."""
    d = mppi_cfg(K, horizon)
    T = orc.mppi_steps(d)
    grp = _group(d, [0] * P)
    one = make_mppi(gpu_pkg, d)
    grp.setWaypoint(*WAYPOINTS[2]); one.setWaypoint(*WAYPOINTS[2])
    u = np.zeros((2, T)); x0 = (0.1, -0.05, 0.3)
    for tick in range(2):
        noise = np.random.default_rng(70 + tick).standard_normal((K, T, 2)) * np.sqrt(0.9)
        ref = orc.mppi_new_controls(d, u, (0, 0), WAYPOINTS[2], x0, noise)
        got = grp.newControls(*x0, noise)
        assert np.allclose(got, ref["out"], rtol=1e-9, atol=1e-12)
        ug = grp.getControls()
        assert np.allclose(ug, ref["u"], rtol=1e-9, atol=1e-12)
        for r in range(P):
            assert np.array_equal(grp.member(r).getControls(), ug)
        u = ref["u"]; grp.setControls(u)
    one.setControls(u)
    for tick in range(3):
        a = grp.newControlsRng(x0, 77, tick)
        b = one.newControlsRng(x0, 77, tick)
        assert np.allclose(a, b, rtol=1e-9, atol=1e-12)
    assert np.allclose(grp.getControls(), one.getControls(), rtol=1e-9, atol=1e-12)
    # a batch enqueued by one call == tick by tick
    g2 = _group(d, [0] * P); g2.setWaypoint(*WAYPOINTS[2]); g2.setControls(grp.getControls())
    grp.enqueueRngBatch(x0, 77, 10, 12); grp.synchronize()
    for i in range(12):
        g2.newControlsRng(x0, 77, 10 + i)
    assert np.array_equal(grp.getControls(), g2.getControls()) and grp.lastControls() == g2.lastControls()
    grp.close(); g2.close(); one.close()


@pytest.mark.parametrize("K,horizon,P", [(2048, 0.5, 2), (8 * 1024, 0.5, 8), (3 * 40000, 0.2, 3)])
def test_mppi_group_direct_exchange_equals_the_all_gather_bit_for_bit(gpu_pkg, K, horizon, P):
    """The members of a one-process group exchange their records DIRECTLY by default (tbnav_mppi_exchange_kind 2: every member stores
    tagged words into every member's buffer, the combine polls its own) — the same kernels as between processes; with
    TBNAV_MPPI_OPT_DIRECT_EXCHANGE 0 the group's all-gather carries them (kind 1).  Same records either way: identical controls,
    tick by tick and through a batch, with the fused small-K kernels (the fold publishes) and the streaming ones (a publish launch)."""
    d = mppi_cfg(K, horizon)
    a, b = _group(d, [0] * P), _group(d, [0] * P)
    b.setOption(8, 0)
    assert all(a.member(r).exchangeKind() == 2 for r in range(P)) and all(b.member(r).exchangeKind() == 1 for r in range(P))
    x0 = (0.05, 0.02, -0.2)
    for g in (a, b):
        g.setWaypoint(*WAYPOINTS[3])
    for tick in range(3):
        assert a.newControlsRng(x0, 31, tick) == b.newControlsRng(x0, 31, tick)
    a.enqueueRngBatch(x0, 31, 10, 9); b.enqueueRngBatch(x0, 31, 10, 9)
    a.synchronize(); b.synchronize()
    assert np.array_equal(a.getControls(), b.getControls()) and a.lastControls() == b.lastControls()
    for r in range(P):
        assert np.array_equal(a.member(r).getControls(), a.getControls())
    noise = np.random.default_rng(3).standard_normal((K, orc.mppi_steps(d), 2)) * np.sqrt(0.9)
    assert a.newControls(*x0, noise) == b.newControls(*x0, noise)
    b.setOption(8, 1)   # and back
    assert all(b.member(r).exchangeKind() == 2 for r in range(P))
    assert a.newControlsRng(x0, 31, 40) == b.newControlsRng(x0, 31, 40)
    a.close(); b.close()


def test_group_rejects_an_ensemble_that_does_not_split_evenly(gpu_pkg):
    from rtn_amd import capi
    with pytest.raises(capi.TbnavError) as ei:
        _group(mppi_cfg(1000, 0.25), [0, 0, 0])
    assert ei.value.status == capi.ERR_INVALID_ARG


# ---- RBPF: the sharded scan issued by the library --------------------------------------------------------------------
def _pf(N, k, **kw):
    from rtn_amd.rbpf import ParticleFilter, default_params
    return ParticleFilter(default_params(N=N, k=k, **kw))


def _pf_group(N, k, devices, pool=0, **kw):
    from rtn_amd.rbpf import ParticleFilterGroup, default_params
    return ParticleFilterGroup(default_params(N=N, k=k, **kw), devices, pool_bytes_per_member=pool)


def _same_state(grp, pf, particles=None):
    pose, prev, w = pf.particles()
    gp, gv, gw = grp.particles()
    assert np.array_equal(gp, pose) and np.array_equal(gv, prev) and np.array_equal(gw, w)
    for p in (range(pf.N) if particles is None else particles):
        assert np.array_equal(grp.logOdds(p), pf.logOdds(p)), p


@pytest.mark.parametrize("world,n_local,heavy", [(2, 6, {3: 0.6, 10: 0.25}), (2, 8, {2: 0.35, 5: 0.55}), (4, 5, {1: 0.6, 17: 0.3}),
                                                 (3, 7, {0: 0.3, 9: 0.3, 20: 0.3}), (3, 4, {11: 0.9}), (4, 3, {5: 0.45, 6: 0.45})])
def test_rbpf_group_equals_the_unsharded_filter_bit_exact(gpu_pkg, world, n_local, heavy):
    """tbnav_rbpf_group with `world` members on device 0: four scans in parity mode (the ensemble's host-drawn normals), a forced
    cross-member resample before the second — the layouts of tests/test_sharded_gpu.py (an exported parent whose own slot is
    taken over, children spanning three members, members whose every slot is imported) — bit-identical to ONE handle holding
    all the particles: Neff, decision, poses, weights, every map.  The exchange is the library's: the weights' all-gather and
    the global normalise on the second stream beside the map update, sizes all-gather, batched export / P2P / import."""
    from dist_workers import rbpf_scenario
    N, k = world * n_local, 8
    grp, pf = _pf_group(N, k, [0] * world), _pf(N, k)
    steps, scans = rbpf_scenario(4)
    resampled = 0
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        normals = orc.normal_stream(50 + s, N * (3 * k + 3) + 1, 0.0, 1.0)
        if s == 1:
            w = np.full(N, 0.01)
            for i, v in heavy.items():
                w[i] = v
            w /= w.sum()
            pf.setParticles(w=w); grp.setParticles(w=w)
        a = grp.SLAM(scans[s], u, cur, prev, True, t_icp, normals)
        b = pf.SLAM(scans[s], u, cur, prev, True, t_icp, normals)
        assert (a.neff, a.resampled, a.n_valid_beams) == (b.neff, b.resampled, b.n_valid_beams) and (a.sum_w, a.sq_sum) == (b.sum_w, b.sq_sum)
        resampled += a.resampled
        _same_state(grp, pf)
    assert resampled >= 1
    assert grp.getRobotState() == pf.getRobotState() and np.array_equal(grp.newMap(), pf.newMap())
    grp.close(); pf.close()


def test_rbpf_group_with_device_noise_draws_what_the_unsharded_filter_draws(gpu_pkg):
    """Production mode: no host normals.  Every member draws ITS slice of the ensemble's Philox stream and the ensemble's resampling
    offset (tbnav_rbpf_set_rng_shard, set by the group), so the group equals the unsharded filter with the same seed bit for bit
    — through a resample the run reaches by itself or by a forced skew — and two members never share a normal."""
    from dist_workers import rbpf_scenario
    world, n_local, k = 4, 6, 8
    N = world * n_local
    grp, pf = _pf_group(N, k, [0] * world), _pf(N, k)
    grp.setSeed(99); pf.setSeed(99)
    steps, scans = rbpf_scenario(4)
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        if s == 2:
            w = np.full(N, 0.01); w[4] = 0.5; w[N - 3] = 0.3; w /= w.sum()
            pf.setParticles(w=w); grp.setParticles(w=w)
        a = grp.SLAM(scans[s], u, cur, prev, True, t_icp, None)
        b = pf.SLAM(scans[s], u, cur, prev, True, t_icp, None)
        assert (a.neff, a.resampled) == (b.neff, b.resampled)
        stride = 3 * k + 3
        full = pf.lastNormals(N * stride + 1)
        for r in range(world):
            mine = grp.member(r).lastNormals(n_local * stride + 1)
            assert np.array_equal(mine[:-1], full[r * n_local * stride:(r + 1) * n_local * stride]) and mine[-1] == full[-1]
        _same_state(grp, pf)
    grp.close(); pf.close()


def test_rccl_communicator_of_one_rank_carries_the_rbpf_scan(gpu_pkg):
    """A REAL RCCL communicator (one rank): the handle's SLAM() is the library's sharded scan — ncclAllGather of the weights on the
    second stream, the global normalise beside the map update, the resample as local table copies — and equals the plain handle
    bit for bit, resample included."""
    from dist_workers import rbpf_scenario
    from rtn_amd.comm import Comm
    comm = Comm.create(Comm.unique_id(), 1, 0, 0)
    N, k = 24, 8
    a, b = _pf(N, k), _pf(N, k)
    a.attachComm(comm)
    steps, scans = rbpf_scenario(4)
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        normals = orc.normal_stream(80 + s, N * (3 * k + 3) + 1, 0.0, 1.0)
        if s == 1:
            w = np.full(N, 0.01); w[3] = 0.6; w[20] = 0.2; w /= w.sum()
            a.setParticles(w=w); b.setParticles(w=w)
        sa = a.SLAM(scans[s], u, cur, prev, True, t_icp, normals)
        sb = b.SLAM(scans[s], u, cur, prev, True, t_icp, normals)
        assert (sa.neff, sa.resampled, sa.sum_w, sa.sq_sum) == (sb.neff, sb.resampled, sb.sum_w, sb.sq_sum)
        for x, y in zip(a.particles(), b.particles()):
            assert np.array_equal(x, y)
        for p in range(N):
            assert np.array_equal(a.logOdds(p), b.logOdds(p))
    a.close(); b.close(); comm.close()


def test_configs4_as_written_100k_particles_8_shards_2000x2000_1080_beams(gpu_pkg):
    """BASELINE configs[4] AS WRITTEN on the one GPU of this box: 100 000 particles, k = 50, 1080-beam scans, a 2000 x 2000 grid,
    split over 8 members of 12 500 (tbnav_rbpf_group, every member on device 0: the exchange is the library's own code path with
    the copy transport; on 8 devices the same calls are ncclAllGather / ncclSend / ncclRecv) — four scans, device noise, a skewed
    weight vector before the third that makes thousands of particles cross member boundaries (one heavy particle's children
    fill three members) — against ONE unsharded handle of 100 000 particles on the same device: Neff, the decision, every pose
    and weight bit for bit, maps of spot particles on every member bit for bit, and those maps against the oracle's GridMapper
    fed each spot particle's lineage of poses."""
    import time
    N, P, k, bd, n_scans = 100_000, 8, 50, 1.0 / 3.0, 4
    nl = N // P
    kw = dict(map_min=-50.0, map_max=50.0, beam_delta_deg=bd)
    from rtn_amd.rbpf import ParticleFilter, default_params
    pf = ParticleFilter(default_params(N=N, k=k, **kw), pool_bytes=80 << 30)
    grp = _pf_group(N, k, [0] * P, pool=10 << 30, **kw)
    assert (pf.xsize, pf.ysize) == (2000, 2000) and grp.n_local == nl
    pf.setSeed(4242); grp.setSeed(4242)
    import rbpf_cases as rc
    steps, poses = rc.trajectory(n_scans, inc=(0.05, 0.04, 0.03))
    rng = np.random.default_rng(8)
    scans = [orc.room_scan(poses[s], n_beams=1080, beam_delta_deg=bd, walls=rc.ROOM_SURVEY, rng=rng) for s in range(n_scans)]
    hist, parents, t_grp, t_one = [], None, [], []
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        if s == 2:
            w = np.full(N, 0.3 / N); w[7] += 0.3; w[60_000] += 0.3; w[N - 1] += 0.1
            pf.setParticles(w=w); grp.setParticles(w=w)
        t0 = time.perf_counter(); a = grp.SLAM(scans[s], u, cur, prev, True, t_icp, None); t_grp.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); b = pf.SLAM(scans[s], u, cur, prev, True, t_icp, None); t_one.append(time.perf_counter() - t0)
        assert a.status == 0 and b.status == 0 and a.n_valid_beams > 600
        assert (a.neff, a.resampled, a.sum_w, a.sq_sum) == (b.neff, b.resampled, b.sum_w, b.sq_sum), s
        hist.append(pf.trace()["new_pose"].copy())
        if s == 2:
            assert b.resampled == 1
            parents = pf.trace()["resample_idx"].copy()
        for x, y in zip(grp.particles(), pf.particles()):
            assert np.array_equal(x, y), s
    print(f"\n[configs4 as written] ms per scan, 8 members on one device / one 100k handle: {[round(t * 1e3, 1) for t in t_grp]} / {[round(t * 1e3, 1) for t in t_one]}")
    kids7 = np.nonzero(parents == 7)[0]
    assert len({int(m) // nl for m in kids7}) >= 3 and len(kids7) > 25_000          # one blob to several members, thousands of slots
    assert int(np.sum(parents // nl != np.arange(N) // nl)) > 10_000                # particles that changed member
    grid, laser = (0.05, -50.0, 50.0, -50.0, 50.0), orc.lds01_laser(bd)
    spots = [0, nl - 1, nl, 3 * nl + 17, 5 * nl + 4321, N - 1]
    for m in spots:
        lo = grp.logOdds(m)
        assert np.array_equal(lo, pf.logOdds(m)), m
    for m in spots[::2]:
        q = int(parents[m])
        g = orc.GridAPI("orc", grid=grid, laser=laser)
        for sc, po in zip(scans, [hist[0][q], hist[1][q], hist[2][q], hist[3][m]]):
            g.integrate_scan(sc, po, esdf=False)
        assert np.array_equal(grp.logOdds(m), g.dump()["log_odds"]), m
        g.close()
    assert grp.getRobotState() == pf.getRobotState()
    grp.close(); pf.close()
