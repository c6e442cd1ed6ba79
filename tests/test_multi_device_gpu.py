"""Everything the library does BETWEEN devices, on two physical MI355X — skipped on a one-GPU box.

The one-GPU suite runs all of this code with every rank / member on device 0 (tests/test_comm_gpu.py, test_comm_processes_gpu.py,
test_host_shims.py): the record layouts, counter spaces, migration plans and status agreements are tested there.  What a single
device cannot show is the part that only exists between devices:
  * tbnav_comm_create_local on distinct devices (ncclCommInitAll) and grouped ncclAllGather / ncclSend / ncclRecv,
  * the direct exchange's stores into a PEER device's fine-grained buffer and a combine that polls words written from another
    device (tbnav_mppi_group: plain pointers + peer access; between processes: hipIpc mappings),
  * tbnav_rbpf_group's cross-device migration, the multi-process RCCL transport, and the class shims' n_gpus on real devices,
  * bench.py under torch.distributed.run as the driver launches it.
The first multi-GPU box that runs `pytest -m gpu` runs these.  The bodies are the one-GPU tests' own (same helpers, same
assertions) with the device list swapped, and `TBNAV_TEST_ALIAS_DEVICES=1` runs them with "two devices" = device 0 twice, so
the test CODE itself is exercised on the one-GPU box (tools/README.md); only the transports differ then."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_api as orc
from cases import WAYPOINTS, make_mppi, mppi_cfg

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALIAS = os.environ.get("TBNAV_TEST_ALIAS_DEVICES") == "1"


@pytest.fixture(scope="module")
def devices(gpu_pkg):
    """[0, 1]: two physical devices; skip on a one-GPU box (unless the alias switch asks for [0, 0])."""
    n = gpu_pkg.capi.lib().tbnav_device_count()
    if n >= 2:
        return [0, 1]
    if ALIAS:
        return [0, 0]
    pytest.skip(f"needs two MI355X ({n} visible); TBNAV_TEST_ALIAS_DEVICES=1 runs the test bodies with device 0 twice")


def _distinct(devices):
    return len(set(devices)) == len(devices)


def _group(d, devices, **kw):
    from rtn_amd.mppi import MPPIGroup, CartModel, LossFunc
    return MPPIGroup(CartModel(d["wheel_radius"], d["wheel_base"]), LossFunc(d["Q"], d["R"], d["P1"]), d["lam"], d["max_wheel_vel"],
                     d["ul_var"], d["ur_var"], d["horizon"], d["dt"], d["rollouts"], devices=devices, **kw)


def test_local_communicators_on_distinct_devices(gpu_pkg, devices):
    """tbnav_comm_create_local(devices): one communicator per member, each on its device; distinct devices go through RCCL
    (ncclCommInitAll), members sharing a device through in-process copies."""
    from rtn_amd.comm import Comm
    comms = Comm.create_local(devices)
    assert [c.rank for c in comms] == [0, 1] and all(c.size == 2 for c in comms)
    assert [c.device for c in comms] == devices
    assert all(c.uses_rccl == _distinct(devices) for c in comms)
    for c in comms:
        c.close()


@pytest.mark.parametrize("K,horizon", [(2048, 0.5), (2 * 40000, 0.2)])
def test_mppi_group_on_two_devices_against_the_oracle_both_exchanges(gpu_pkg, devices, K, horizon):
    """tbnav_mppi_group over two devices — fused small-K kernels (the fold publishes) and the streaming ones (a publish launch):
    host-noise ticks against the oracle over the whole ensemble with the direct exchange (kind 2: each member's kernels store
    tagged words into the OTHER device's buffer, the combine polls its own) and with the group's all-gather (kind 1: grouped
    ncclAllGather between distinct devices); the two end in identical bits, tick by tick and through a batch; every member's
    warm start identical; production ticks equal ONE handle holding the ensemble (the members draw disjoint slices of one
    counter space)."""
    d = mppi_cfg(K, horizon)
    T = orc.mppi_steps(d)
    a, b = _group(d, devices), _group(d, devices)
    b.setOption(8, 0)   # TBNAV_MPPI_OPT_DIRECT_EXCHANGE
    assert all(a.member(r).exchangeKind() == 2 for r in range(2)), "the direct exchange's self-test failed between these devices"
    assert all(b.member(r).exchangeKind() == 1 for r in range(2))
    x0 = (0.1, -0.05, 0.3)
    for g in (a, b):
        g.setWaypoint(*WAYPOINTS[2])
    u = np.zeros((2, T))
    for tick in range(2):
        noise = np.random.default_rng(70 + tick).standard_normal((K, T, 2)) * np.sqrt(0.9)
        ref = orc.mppi_new_controls(d, u, (0, 0), WAYPOINTS[2], x0, noise)
        ga, gb = a.newControls(*x0, noise), b.newControls(*x0, noise)
        assert ga == gb
        assert np.allclose(ga, ref["out"], rtol=1e-9, atol=1e-12) and np.allclose(a.getControls(), ref["u"], rtol=1e-9, atol=1e-12)
        assert np.array_equal(a.getControls(), b.getControls())
        for r in range(2):
            assert np.array_equal(a.member(r).getControls(), a.getControls()) and np.array_equal(b.member(r).getControls(), a.getControls())
        u = ref["u"]
        a.setControls(u); b.setControls(u)
    one = make_mppi(gpu_pkg, d, device=devices[0])
    one.setWaypoint(*WAYPOINTS[2]); one.setControls(u)
    for tick in range(3):
        ra, rb = a.newControlsRng(x0, 77, tick), b.newControlsRng(x0, 77, tick)
        assert ra == rb and np.allclose(ra, one.newControlsRng(x0, 77, tick), rtol=1e-9, atol=1e-12)
    a.enqueueRngBatch(x0, 77, 10, 40); b.enqueueRngBatch(x0, 77, 10, 40)
    a.synchronize(); b.synchronize()
    assert np.array_equal(a.getControls(), b.getControls()) and a.lastControls() == b.lastControls()
    a.close(); b.close(); one.close()


def test_mppi_group_reattached_exchange_starts_from_a_clean_buffer(gpu_pkg, devices):
    """Switching the direct exchange off and on again re-runs its set-up: fresh zeroed buffers, sequence numbers from 1 on every
    member (a tag can never equal the zeroed buffer's 0, and nothing of the previous attachment's words survives).  The ticks
    after the switch equal the same ticks of a group that never switched."""
    d = mppi_cfg(2048, 0.5)
    a, b = _group(d, devices), _group(d, devices)
    x0 = (0.05, 0.02, -0.2)
    for g in (a, b):
        g.setWaypoint(*WAYPOINTS[3])
    for tick in range(5):   # an ODD number of ticks: the parity buffers of a and b differ from here on unless the set-up resets them
        assert a.newControlsRng(x0, 31, tick) == b.newControlsRng(x0, 31, tick)
    a.setOption(8, 0)
    assert all(a.member(r).exchangeKind() == 1 for r in range(2))
    assert a.newControlsRng(x0, 31, 5) == b.newControlsRng(x0, 31, 5)
    a.setOption(8, 1)
    assert all(a.member(r).exchangeKind() == 2 for r in range(2))
    for tick in range(6, 12):
        assert a.newControlsRng(x0, 31, tick) == b.newControlsRng(x0, 31, tick)
    assert np.array_equal(a.getControls(), b.getControls())
    a.close(); b.close()


@pytest.mark.parametrize("n_local,heavy,device_noise", [(6, {3: 0.6, 10: 0.25}, False), (8, {2: 0.35, 5: 0.55}, False), (7, {1: 0.9}, True)])
def test_rbpf_group_on_two_devices_equals_one_handle_through_a_cross_device_resample(gpu_pkg, devices, n_local, heavy, device_noise):
    """tbnav_rbpf_group over two devices against ONE handle holding all particles, bit for bit: four scans, a forced resample
    before the second whose heavy particles send children ACROSS the device boundary (tile blobs over ncclSend / ncclRecv), the
    weights' all-gather and the global normalise on the second stream of each device.  Parity mode (host normals) and device
    noise (each member draws its slice of the ensemble's stream)."""
    from dist_workers import rbpf_scenario
    from rtn_amd.rbpf import ParticleFilter, ParticleFilterGroup, default_params
    N, k = 2 * n_local, 8
    grp = ParticleFilterGroup(default_params(N=N, k=k), devices)
    pf = ParticleFilter(default_params(N=N, k=k, device=devices[0]))
    if device_noise:
        grp.setSeed(99); pf.setSeed(99)
    steps, scans = rbpf_scenario(4)
    resampled = 0
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        normals = None if device_noise else orc.normal_stream(50 + s, N * (3 * k + 3) + 1, 0.0, 1.0)
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
        for x, y in zip(grp.particles(), pf.particles()):
            assert np.array_equal(x, y), s
        for p in range(N):
            assert np.array_equal(grp.logOdds(p), pf.logOdds(p)), (s, p)
    assert resampled >= 1
    assert grp.getRobotState() == pf.getRobotState() and np.array_equal(grp.newMap(), pf.newMap())
    grp.close(); pf.close()


def _transport(devices):
    return "rccl" if _distinct(devices) else "ipc"


def test_two_processes_over_the_rccl_transport(gpu_pkg, devices):
    """One PROCESS per device, the library's own RCCL communicator between them (ncclCommInitRank with the id carried by the
    job's process group): the transport's self-test, the sharded MPPI tick — direct exchange through hipIpc-mapped buffers of
    the PEER DEVICE, and the all-gather — and the sharded RBPF scan with cross-rank migration; every worker asserts against the
    oracle / the unsharded handle as in tests/test_comm_processes_gpu.py."""
    from dist_workers import comm_selftest_worker, mppi_comm_worker, rbpf_comm_worker, run_spawn
    tr = _transport(devices)
    out = run_spawn(comm_selftest_worker, 2, 1 << 18, transport=tr)
    assert all(out[r] is True for r in range(2))
    for K_local, horizon in ((1024, 0.5), (40000, 0.24)):
        direct = run_spawn(mppi_comm_worker, 2, K_local, horizon, 3, transport=tr)
        via = run_spawn(mppi_comm_worker, 2, K_local, horizon, 3, False, transport=tr)
        assert all(direct[r]["kind"] == 2 for r in range(2)), "the direct exchange's self-test failed between these devices"
        assert all(via[r]["kind"] == 1 for r in range(2))
        for o in (direct, via):
            for (g0, u0), (g1, u1) in zip(o[0]["host"], o[1]["host"]):
                assert np.array_equal(g0, g1) and np.array_equal(u0, u1)
            assert all(np.array_equal(x, y) for x, y in zip(o[0]["rng"], o[1]["rng"])) and np.array_equal(o[0]["batch_u"], o[1]["batch_u"])
        assert all(np.array_equal(x, y) for x, y in zip(direct[0]["rng"], via[0]["rng"])) and np.array_equal(direct[0]["batch_u"], via[0]["batch_u"])
    for n_local, heavy, device_noise in ((6, {3: 0.6, 10: 0.25}, False), (7, {1: 0.9}, True)):
        out = run_spawn(rbpf_comm_worker, 2, n_local, 8, heavy, device_noise, transport=tr)
        assert out[0]["resampled"] >= 1 and out[1]["stats"] == out[0]["stats"]


def test_class_shims_with_n_gpus_on_two_devices(gpu_pkg, devices):
    """controller::MPPI(..., n_gpus = 2) and bmapping::ParticleFilter(..., n_gpus = 2) with their members on devices 0 and 1
    (host/src/test_hooks.cpp): the MPPI class against the oracle fed the seeded host twister's stream, the filter class identical
    to itself on one GPU."""
    import rbpf_cases as rc
    import test_host_shims as ths
    host = C.CDLL(ths.HOST_LIB)
    host.hst_last_error.restype = C.c_char_p
    host.hst_distinct_devices(1 if _distinct(devices) else 0)
    try:
        d = dict(ths.MPPI_BASE, rollouts=256)
        T, K, n_ticks = 25, 256, 3
        out = np.empty(2 * n_ticks); u_dev = np.empty((2, T))
        host.hst_mppi_gpus(2)
        try:
            got_T = host.hst_mppi_tick(ths._p(ths._mppi_params(d)), K, C.c_uint64(9), ths._p(ths._arr(WAYPOINTS[1])), ths._p(ths._arr([0.0, 0.0, 0.0])),
                                       n_ticks, ths._p(out), ths._p(u_dev))
        finally:
            host.hst_mppi_gpus(1)
        assert got_T == T, host.hst_last_error()
        stream = orc.normal_stream(9, n_ticks * K * T * 2, 0.0, np.sqrt(0.9)).reshape(n_ticks, K, T, 2)
        u = np.zeros((2, T))
        for t in range(n_ticks):
            ref = orc.mppi_new_controls(d, u, (0, 0), WAYPOINTS[1], (0, 0, 0), stream[t])
            u = ref["u"]
            assert np.allclose(out[2 * t:2 * t + 2], ref["out"], rtol=1e-9, atol=1e-12)
        assert np.allclose(u_dev, u, rtol=1e-9, atol=1e-12)

        N, k, n_scans = 40, 50, 6
        steps, poses = rc.trajectory(n_scans, inc=(0.04, 0.03, 0.02))
        rng = np.random.default_rng(3)
        scans = np.stack([orc.room_scan(poses[s], walls=rc.ROOM_SMALL, rng=rng) for s in range(n_scans)])
        odom = np.stack([steps[0][0]] + [st[1] for st in steps])
        res = []
        for gpus in (1, 2):
            out_pose = np.empty((n_scans, 3)); out_neff = np.empty(n_scans, dtype=np.int32); m = np.empty(80 * 80, dtype=np.int8)
            host.hst_pf_reference_field(0); host.hst_pf_gpus(gpus)
            try:
                xs = host.hst_pf_run(N, k, C.c_double(2.0), C.c_uint64(11), ths._p(scans), 360, n_scans, ths._p(odom), ths._p(out_pose), ths._p(out_neff), ths._p(m))
            finally:
                host.hst_pf_reference_field(1); host.hst_pf_gpus(1)
            assert xs == 80, host.hst_last_error()
            res.append((out_pose, out_neff, m))
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    finally:
        host.hst_distinct_devices(0)


def test_bench_two_gpus_as_the_driver_launches_it(gpu_pkg, devices):
    """bench.py --gpus 2 under torch.distributed.run, one rank per device over RCCL — no dev switch: the line must say which
    exchange carried the headline (`exchange`: DIRECT, or the all-gather if the direct exchange's self-test failed on this node —
    visibly, not silently), time the same tick through the all-gather beside it, and carry both multi-GPU legs."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    if not _distinct(devices):
        env["TBNAV_BENCH_ONE_GPU_GLOO"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert "tbnav_mppi_attach_comm" in line["exchange"], line["exchange"]
    assert line["exchange"].startswith("direct") or "ncclAllGather" in line["exchange"] or "all-gather" in line["exchange"]
    assert "multi_gpu_legs" not in line, line.get("multi_gpu_legs")
    assert line["weak_tick_via_comm_all_gather"]["exchange_kind"] == 1 and line["weak_tick_via_comm_all_gather"]["rollouts_per_s"] > 0
    assert line["strong_scaling_configs3"]["rollouts_per_s"] > 0
    assert line["rbpf_sharded"]["resamples"] >= 2 and "tbnav_rbpf_attach_comm" in line["rbpf_sharded"]["exchange"]
    if _distinct(devices):
        assert "RCCL" in line["rbpf_sharded"]["exchange"]
