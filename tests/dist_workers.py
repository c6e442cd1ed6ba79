"""Worker entry points for the world_size-2 tests (spawned processes; gloo over 127.0.0.1)."""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _init(rank, world, port, backend="gloo"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        import torch
        torch.cuda.set_device(rank)
    dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def run_spawn(fn, world, *args, backend="gloo", transport="ipc"):
    """Spawn `world` processes running fn(rank, world, *args) under a `backend` process group; returns {rank: result}.
    transport: what _ipc_comm() builds in the workers ("ipc": all ranks on device 0; "rccl": rank r on device r)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    res = mgr.dict()
    procs = [ctx.Process(target=_guard, args=(fn, r, world, port, res, backend, transport) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    for p in procs:
        if p.is_alive():
            p.kill()
    out = dict(res)
    errs = [v for k, v in out.items() if isinstance(k, str) and k.startswith("err")]
    assert not errs, "\n".join(errs)
    assert all(r in out for r in range(world)), f"missing ranks: {sorted(k for k in out)}"
    return out


def _guard(fn, rank, world, port, res, backend, transport, *args):
    global _TRANSPORT
    _TRANSPORT = transport
    try:
        dist = _init(rank, world, port, backend)
        res[rank] = fn(rank, world, *args)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        res[f"err{rank}"] = f"rank {rank}:\n{traceback.format_exc()}"


# ---- MPPI: oracle compute per rank, real exchange -------------------------------------------------
class OracleMppiBackend:
    """Per-rank compute by the oracle (CPU): same partial-record contract as the HIP handle."""

    def __init__(self, cfg_local, cfg_global, xd, uinit):
        import oracle_api as orc
        self.orc, self.cl, self.cg, self.xd, self.uinit = orc, cfg_local, cfg_global, xd, uinit
        self.T = orc.mppi_steps(cfg_local)
        self.u = np.zeros((2, self.T))
        self.out = None

    def partials(self, x0, noise):
        import torch
        _, rec = self.orc.mppi_shard_partials(self.cl, self.u, self.xd, x0, noise)
        return torch.from_numpy(rec.reshape(self.T, 1, 8).copy())

    def combine(self, records_all, n_shards):
        rec = records_all.numpy().reshape(n_shards, self.T, 8)
        self.u, self.out = self.orc.mppi_combine(self.cg, self.u, self.uinit, rec)

    def result(self):
        return self.out


def mppi_worker(rank, world, K_global, horizon, n_ticks, seed):
    import __graft_entry__ as g
    g.load_package()
    import oracle_api as orc
    from cases import WAYPOINTS, mppi_cfg
    from rtn_amd.sharded import ShardedMPPI
    Kl = K_global // world
    cg, cl = mppi_cfg(K_global, horizon), mppi_cfg(Kl, horizon)
    T = orc.mppi_steps(cg)
    sm = ShardedMPPI(OracleMppiBackend(cl, cg, WAYPOINTS[1], (0.0, 0.0)))
    outs = []
    for t in range(n_ticks):
        noise = orc.normal_stream(seed + t, K_global * T * 2, 0.0, np.sqrt(0.9)).reshape(K_global, T, 2)
        sm.tick((0.0, 0.0, 0.0), noise[rank * Kl:(rank + 1) * Kl])
        outs.append(sm.result())
    return dict(outs=outs, u=sm.b.u)


# ---- RBPF: fake per-rank compute, real exchange + the product's resample_global -------------------
class FakeRbpfBackend:
    """Stands in for the HIP handle: deterministic per-particle 'update', a small vector as the map; blobs of
    DIFFERENT sizes per particle (like tiled maps), so the size exchange is exercised."""

    def __init__(self, rank, n_local, map_len=32):
        import torch
        self.torch = torch
        self.device = torch.device("cpu")
        self.n_local, self.rank, self.map_len = n_local, rank, map_len
        gid = np.arange(rank * n_local, (rank + 1) * n_local)
        self.state = np.stack([gid + 0.25, gid * 2.0, gid * 3.0, gid + 0.5, gid * 5.0, gid * 7.0, np.zeros(n_local)], 1)
        self.maps = gid[:, None] * 1000.0 + np.arange(map_len)[None, :]
        self.dist_maps = -self.maps
        self.pad = gid % 3  # extra doubles in the blob

    def slam_local(self, scan, u, cur, prev, icp_ok, T_icp, normals_local):
        self.state[:, 6] = np.abs(normals_local[:self.n_local]) ** 4 + 1e-3  # raw weights from this rank's draws
        return None

    def weights_tensor(self):
        return self.torch.from_numpy(self.state[:, 6].copy())

    def resample(self, w_all, offset, z):
        from rtn_amd.rbpf import resample_global
        parents, wn, st = resample_global(w_all.numpy(), z)
        self._wn = wn
        self.state[:, 6] = wn[offset:offset + self.n_local]
        return st, (parents if st.resampled else None)

    def set_weights_after_resample(self, gp):
        self.state[:, 6] = self._wn[gp]

    def _blob(self, slot):
        return np.concatenate([self.state[slot], self.maps[slot], self.dist_maps[slot], [float(self.pad[slot])], np.zeros(int(self.pad[slot]))])

    def export_batch(self, slots):
        blobs = [self._blob(sl) for sl in slots]
        offs = np.zeros(len(slots) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([8 * b.size for b in blobs])
        if not blobs:
            return self.torch.empty(0, dtype=self.torch.uint8), offs
        return self.torch.from_numpy(np.concatenate(blobs).copy()).view(self.torch.uint8), offs

    def new_blob(self, n):
        return self.torch.empty(n, dtype=self.torch.uint8)

    def import_batch(self, slots, buf, offsets):
        v = buf.view(self.torch.float64).numpy()
        L = self.map_len
        for slot, off in zip(slots, offsets):
            w = v[off // 8:]
            self.state[slot], self.maps[slot], self.dist_maps[slot] = w[:7], w[7:7 + L], w[7 + L:7 + 2 * L]
            self.pad[slot] = int(w[7 + 2 * L])

    def gather_local(self, local_parent):
        src = np.where(local_parent < 0, np.arange(self.n_local), local_parent)
        self.state, self.maps, self.dist_maps, self.pad = self.state[src].copy(), self.maps[src].copy(), self.dist_maps[src].copy(), self.pad[src].copy()


def rbpf_worker(rank, world, n_local, seed):
    import __graft_entry__ as g
    g.load_package()
    from rtn_amd.sharded import ShardedRBPF
    N = n_local * world
    b = FakeRbpfBackend(rank, n_local)
    sr = ShardedRBPF(b)
    rng = np.random.default_rng(seed)
    normals = rng.standard_normal(N * 1 + 1)
    st, _, parents = sr.tick(None, None, None, None, True, None, normals, 1)
    return dict(state=b.state, maps=b.maps, dist=b.dist_maps, parents=parents, neff=st.neff, resampled=st.resampled)


# ---- HIP compute per rank (two processes sharing ONE GPU, gloo exchange) -----------------------------
def mppi_hip_worker(rank, world, K_global, horizon, n_ticks, seed):
    import torch
    import __graft_entry__ as g
    g.load_package()
    import oracle_api as orc
    from cases import WAYPOINTS, make_mppi, mppi_cfg
    from rtn_amd.sharded import HipShardBackend, ShardedMPPI
    Kl = K_global // world
    T = orc.mppi_steps(mppi_cfg(K_global, horizon))
    m = make_mppi(None, mppi_cfg(Kl, horizon), 0)
    m.setWaypoint(*WAYPOINTS[1])
    dev = torch.device("cuda", 0)
    sm = ShardedMPPI(HipShardBackend(m, dev))
    outs = []
    for t in range(n_ticks):
        noise = orc.normal_stream(seed + t, K_global * T * 2, 0.0, np.sqrt(0.9)).reshape(K_global, T, 2)
        nz = torch.from_numpy(noise[rank * Kl:(rank + 1) * Kl]).to(dev)
        a, b = nz[:, :, 0].t().contiguous(), nz[:, :, 1].t().contiguous()
        sm.tick((0.0, 0.0, 0.0), (a.data_ptr(), b.data_ptr()))
        outs.append(sm.result())
    return dict(outs=outs, u=m.getControls())


def rbpf_scenario(n_scans=3):
    import oracle_api as orc
    import rbpf_cases as rc
    steps, poses = rc.trajectory(n_scans, inc=(0.03, 0.02, 0.02))
    rng = np.random.default_rng(5)
    scans = [orc.room_scan(poses[s], walls=rc.ROOM_SMALL, rng=rng) for s in range(n_scans)]
    return steps, scans


def rbpf_hip_worker(rank, world, n_local, k, skew_scan, heavy=None, df_inject=False, backend="gloo"):
    """heavy: {global index: weight} forced before scan `skew_scan` (default: 3 -> 0.6, N-2 -> 0.25)."""
    import torch
    import __graft_entry__ as g
    g.load_package()
    import oracle_api as orc
    from rtn_amd.rbpf import ParticleFilter, default_params
    from rtn_amd.sharded import HipRbpfShardBackend, ShardedRBPF
    N = n_local * world
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    pf = ParticleFilter(default_params(N=n_local, k=k, device=dev.index))
    sr = ShardedRBPF(HipRbpfShardBackend(pf, dev))
    steps, scans = rbpf_scenario(4)
    stride = 3 * k + 3
    hist = []
    heavy = heavy or {3: 0.6, N - 2: 0.25}
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        normals = orc.normal_stream(50 + s, N * stride + 1, 0.0, 1.0)
        if s == skew_scan:
            w = np.full(N, 0.01)
            for i, v in heavy.items():
                w[i] = v
            w /= w.sum()
            pf.setParticles(w=w[rank * n_local:(rank + 1) * n_local])
        st, _, parents = sr.tick(scans[s], u, cur, prev, True, t_icp, normals, stride)
        hist.append((st.neff, st.resampled, parents.tolist()))
    pose, prev_pose, w = pf.particles()
    return dict(pose=pose, prev=prev_pose, w=w, hist=hist, lo=[pf.logOdds(p) for p in range(n_local)],
                codes=[pf.distCode(p) for p in range(n_local)], migrated=sr.bytes_migrated)


# ---- the in-library sharded paths with one PROCESS per rank (include/tbnav_comm.h) ------------------------------------------
# On a one-GPU box every rank sits on device 0, which RCCL refuses: the communicator is made on the IPC transport
# (tbnav_comm_unique_id_ipc; rank 0's id travels over this job's gloo group).  Everything above the transport — attach, the
# rank's offsets into the gather buffers and the noise counter space, the status agreement, the migration plan — is the code
# the RCCL job runs.
_TRANSPORT = "ipc"   # "rccl": one DEVICE per rank (rank r on device r) over the library's RCCL communicator — tests/test_multi_device_gpu.py


def _ipc_comm():
    """This rank's communicator: the IPC transport with every rank on device 0 (the one-GPU box), or — when the spawning test
    set transport="rccl" — RCCL with rank r on device r."""
    import __graft_entry__ as g
    g.load_package()
    import torch
    import torch.distributed as dist
    from rtn_amd.comm import Comm
    if _TRANSPORT == "rccl":
        dev = dist.get_rank()
        torch.cuda.set_device(dev)
        comm = Comm.from_torch_distributed(dev, transport="rccl")
        assert comm.uses_rccl is True and comm.device == dev
        return comm
    torch.cuda.set_device(0)
    comm = Comm.from_torch_distributed(0, transport="ipc")
    assert comm.uses_rccl is False
    return comm


def comm_selftest_worker(rank, world, nbytes):
    comm = _ipc_comm()
    assert (comm.rank, comm.size, comm.device) == (rank, world, rank if _TRANSPORT == "rccl" else 0)
    comm.selftest(nbytes)   # one all-gather and one ring of point-to-point messages, checked on every rank
    comm.close()
    return True


def mppi_comm_worker(rank, world, K_local, horizon, n_ticks, direct=True):
    """This rank's shard of a K_local * world ensemble behind an attached handle: host-noise ticks (its slice of the ensemble's
    perturbations), production ticks one by one and as one batch; returns what the parent compares across ranks and against
    the oracle / an unsharded handle."""
    import torch
    comm = _ipc_comm()
    import oracle_api as orc
    from cases import WAYPOINTS, make_mppi, mppi_cfg
    import __graft_entry__ as g
    pkg = g.load_package()
    K = K_local * world
    d, dl = mppi_cfg(K, horizon), mppi_cfg(K_local, horizon)
    T = orc.mppi_steps(d)
    m = make_mppi(pkg, dl)
    m.setWaypoint(*WAYPOINTS[2])
    m.setDirectExchange(direct)
    m.attachComm(comm)
    x0 = (0.1, -0.05, 0.3)
    out = {"host": [], "rng": [], "kind": m.exchangeKind()}
    u = np.zeros((2, T))
    for tick in range(2):
        noise = np.random.default_rng(70 + tick).standard_normal((K, T, 2)) * np.sqrt(0.9)
        got = m.newControls(*x0, np.ascontiguousarray(noise[rank * K_local:(rank + 1) * K_local]))
        out["host"].append((np.array(got), m.getControls().copy()))
        if rank == 0:
            ref = orc.mppi_new_controls(d, u, (0, 0), WAYPOINTS[2], x0, noise)
            assert np.allclose(got, ref["out"], rtol=1e-9, atol=1e-12) and np.allclose(m.getControls(), ref["u"], rtol=1e-9, atol=1e-12)
        u = m.getControls().copy()
    for tick in range(n_ticks):
        out["rng"].append(np.array(m.newControlsRng(x0, 77, tick)))
    st = torch.cuda.Stream()
    m.enqueueRngBatch(x0, 77, 100, 6, st.cuda_stream)
    torch.cuda.synchronize()
    out["batch_u"] = m.getControls().copy()
    out["batch_last"] = np.array(m.lastControls(st.cuda_stream))
    m.close()
    if rank == 0:  # the same production ticks on ONE handle holding the whole ensemble (same counter space: same perturbations)
        one = make_mppi(pkg, d)
        one.setWaypoint(*WAYPOINTS[2]); one.setControls(u)
        for tick in range(n_ticks):
            assert np.allclose(one.newControlsRng(x0, 77, tick), out["rng"][tick], rtol=1e-9, atol=1e-12), tick
        for i in range(6):
            one.newControlsRng(x0, 77, 100 + i)
        assert np.allclose(one.getControls(), out["batch_u"], rtol=1e-9, atol=1e-12)
        one.close()
    comm.close()
    return out


def rbpf_comm_worker(rank, world, n_local, k, heavy, device_noise):
    """This rank's shard of an n_local * world filter behind an attached handle, four scans with a forced cross-rank resample
    before the second; the unsharded filter runs beside it in the same process and every local particle is compared with its
    global twin bit for bit (poses, weights, maps), as are Neff / the decision."""
    comm = _ipc_comm()
    import oracle_api as orc
    from rtn_amd.rbpf import ParticleFilter, default_params
    N = n_local * world
    lo = rank * n_local
    a = ParticleFilter(default_params(N=n_local, k=k))
    b = ParticleFilter(default_params(N=N, k=k))
    a.setParticles(w=np.full(n_local, 1.0 / N))
    if device_noise:
        a.setSeed(99); b.setSeed(99)
    a.attachComm(comm)
    steps, scans = rbpf_scenario(4)
    stride = 3 * k + 3
    resampled, stats = 0, []
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        full = None if device_noise else orc.normal_stream(50 + s, N * stride + 1, 0.0, 1.0)
        mine = None if device_noise else np.concatenate([full[lo * stride:(lo + n_local) * stride], full[-1:]])
        if s == 1:
            w = np.full(N, 0.01)
            for i, v in heavy.items():
                w[i] = v
            w /= w.sum()
            a.setParticles(w=w[lo:lo + n_local]); b.setParticles(w=w)
        sa = a.SLAM(scans[s], u, cur, prev, True, t_icp, mine)
        sb = b.SLAM(scans[s], u, cur, prev, True, t_icp, full)
        assert (sa.neff, sa.resampled, sa.sum_w, sa.sq_sum, sa.n_valid_beams) == (sb.neff, sb.resampled, sb.sum_w, sb.sq_sum, sb.n_valid_beams), s
        resampled += sa.resampled
        for x, y in zip(a.particles(), b.particles()):
            assert np.array_equal(x, y[lo:lo + n_local]), s
        for p in range(n_local):
            assert np.array_equal(a.logOdds(p), b.logOdds(lo + p)), (s, p)
        stats.append((sa.neff, sa.resampled))
    a.close(); b.close(); comm.close()
    return {"resampled": resampled, "stats": stats}


def rbpf_rank_failure_worker(rank, world, n_local, failing_rank):
    """A failure only ONE rank sees — its tile pool runs dry in the map update of the second scan — must stop EVERY rank at that
    scan with that status (the ranks agree on a status word per scan; round-3 advisor finding: a rank that left alone had its
    peers wait in the next collective for ever).  Returns the status of every scan as this rank saw it, and how long it took."""
    import time
    comm = _ipc_comm()
    from rtn_amd.rbpf import ParticleFilter, default_params
    # the failing rank: room for the first scan's tiles of its particles (an 80 x 80 map is 3 x 3 tiles) but not for the second
    # scan's clones after its tables were shared by a forced resample ... simpler and deterministic: a pool of ONE particle's tiles
    pool = 8320 * 12 if rank == failing_rank else 0
    a = ParticleFilter(default_params(N=n_local, k=6), pool_bytes=pool)
    a.setParticles(w=np.full(n_local, 1.0 / (n_local * world)))
    a.setSeed(7)
    a.attachComm(comm)
    steps, scans = rbpf_scenario(3)
    out, t0 = [], time.perf_counter()
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        st = a.SLAM(scans[s], u, cur, prev, True, t_icp, None, check=False)
        out.append(int(st.status))
        if st.status != 0:
            break
    waited = time.perf_counter() - t0
    a.close(); comm.close()
    return {"status": out, "waited": waited}


def mppi_direct_fault_worker(rank, world, K_local, horizon):
    """The direct exchange with a fault injected on every rank (records never reach the peers; bound 0.3 s): a batch of ticks must
    come back with an error status in about one bound — not one bound per tick, not a hang — and the handle must work again, through
    the communicator's all-gather, after a detach / re-attach (what bench.py does when its headline fails that way)."""
    import time
    import torch
    comm = _ipc_comm()
    import __graft_entry__ as g
    pkg = g.load_package()
    from cases import WAYPOINTS, make_mppi, mppi_cfg
    from rtn_amd import capi
    m = make_mppi(pkg, mppi_cfg(K_local, horizon))
    m.setWaypoint(*WAYPOINTS[1])
    m.setDirectExchange(2)
    m.attachComm(comm)
    assert m.exchangeKind() == 2   # (the self-test runs before the fault is switched on)
    st = torch.cuda.Stream()
    t0 = time.perf_counter()
    m.enqueueRngBatch((0.0, 0.0, 0.0), 5, 0, 6, st.cuda_stream)
    msg = None
    try:
        m.lastControls(st.cuda_stream)
    except capi.TbnavError as e:
        msg = str(e)
    waited = time.perf_counter() - t0
    m.attachComm(None)
    m.setDirectExchange(0)
    m.setInitialControls(0.0, 0.0)
    m.attachComm(comm)
    kind = m.exchangeKind()
    out = [np.array(m.newControlsRng((0.0, 0.0, 0.0), 5, i)) for i in range(3)]
    m.close(); comm.close()
    return {"msg": msg, "waited": waited, "kind": kind, "out": out}


def mppi_rank_failure_worker(rank, world, K_local, horizon, failing_rank):
    """The all-gather exchange with ONE rank's own rollouts failing in the third tick (TBNAV_MPPI_OPT_FAULT_INJECT): that rank gets
    the error from its enqueue; every other rank's enqueue succeeds (the rank cannot know yet) and its last_controls / next enqueue
    report the failure; NO rank's controls are updated by that tick — they are the previous controls, shifted — and all ranks'
    warm starts stay identical and finite (round 4 poisoned the combine with NaN records, which the clamp turned into
    u = -max_wheel_vel on the healthy ranks: advisor finding); after a detach / re-attach the ensemble works again."""
    import time
    import torch
    comm = _ipc_comm()
    import __graft_entry__ as g
    pkg = g.load_package()
    from cases import WAYPOINTS, make_mppi, mppi_cfg
    from rtn_amd import capi
    m = make_mppi(pkg, mppi_cfg(K_local, horizon))
    m.setWaypoint(*WAYPOINTS[1])
    m.setDirectExchange(0)
    m.attachComm(comm)
    assert m.exchangeKind() == 1
    st = torch.cuda.Stream()
    for i in range(2):
        m.newControlsRng((0.0, 0.0, 0.0), 5, i, st.cuda_stream)
    before = m.getControls()   # (materialises the owed shift: what tick 2 reads)
    if rank == failing_rank:
        m.setOption(capi.MPPI_OPT_FAULT_INJECT, 1)
    t0 = time.perf_counter()
    enq_msg = last_msg = again_msg = None
    try:
        m.enqueueRng((0.0, 0.0, 0.0), 5, 2, st.cuda_stream)
    except capi.TbnavError as e:
        enq_msg = str(e)
    try:
        m.lastControls(st.cuda_stream)
    except capi.TbnavError as e:
        last_msg = str(e)
    try:   # latched on every rank: each goes on JOINING the all-gather (round 6), with records that say so, and returns the error
        m.enqueueRng((0.0, 0.0, 0.0), 5, 3, st.cuda_stream)
    except capi.TbnavError as e:
        again_msg = str(e)
    waited = time.perf_counter() - t0
    after = m.getControls()
    m.attachComm(None)
    m.setControls(before)
    m.attachComm(comm)
    good = [np.array(m.newControlsRng((0.0, 0.0, 0.0), 5, 2 + i, st.cuda_stream)) for i in range(2)]
    m.close(); comm.close()
    return {"enq": enq_msg, "last": last_msg, "again": again_msg, "waited": waited, "before": before, "after": after, "good": good}


def mppi_soak_worker(rank, world, K_local, horizon, n_ticks, direct):
    """Many production ticks in batches through the attached handle (sequence numbers, buffer parity, the dead mark never raised);
    returns the final controls."""
    import torch
    comm = _ipc_comm()
    import __graft_entry__ as g
    pkg = g.load_package()
    from cases import WAYPOINTS, make_mppi, mppi_cfg
    m = make_mppi(pkg, mppi_cfg(K_local, horizon))
    m.setWaypoint(*WAYPOINTS[1])
    m.setDirectExchange(direct)
    m.attachComm(comm)
    st = torch.cuda.Stream()
    done = 0
    while done < n_ticks:
        n = min(5000, n_ticks - done)
        m.enqueueRngBatch((0.0, 0.0, 0.0), 9, done, n, st.cuda_stream)
        m.lastControls(st.cuda_stream)   # (raises if an exchange ran out of time)
        done += n
    out = {"u": m.getControls().copy(), "kind": m.exchangeKind()}
    m.close(); comm.close()
    return out
