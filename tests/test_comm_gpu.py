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


def test_group_rejects_an_ensemble_that_does_not_split_evenly(gpu_pkg):
    from rtn_amd import capi
    with pytest.raises(capi.TbnavError) as ei:
        _group(mppi_cfg(1000, 0.25), [0, 0, 0])
    assert ei.value.status == capi.ERR_INVALID_ARG
