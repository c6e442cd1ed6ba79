"""Ensemble sharding of the two hot paths across ranks (one process per GPU, torch.distributed).

SURVEY.md section 8-e.  Both paths shard their ensemble (rollouts / particles) with NO data-path
collective on the big arrays; each has exactly one small exchange per update:

MPPI   rollouts are independent until the per-time-step soft-min.  Each rank rolls out its K/P slice
       and emits T partial records (include/tbnav_mppi.h); ONE all-gather of those records (RCCL over
       xGMI on GPUs; 8*T*S doubles, latency-bound) and every rank runs the same combine, so all ranks
       hold the same warm-start controls afterwards.

RBPF   the per-particle update (sample, score, proposal, raycast, distance field) is rank-local.
       ONE all-gather of the raw weights (N doubles); every rank then runs the reference's sequential
       normalise / Neff / low-variance selection on the identical global vector
       (tbnav_rbpf_resample_global), so Neff and the parent list are bit-exact and identical
       everywhere.  Only when resampling fires do particles move: slots whose parent lives on
       another rank receive that parent's state (pose, weight, log-odds, distance codes) in a
       point-to-point exchange — the one bandwidth-bound step (maps are MBs per particle).

The per-rank compute sits behind small backend classes so the exchange logic runs on CPU with the
gloo backend in tests (tests/test_sharded_gloo.py plugs the oracle / a fake in) and on GPUs with the
HIP handles.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import capi


def _world(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def _backend_is_gloo(group=None):
    return dist.is_initialized() and dist.get_backend(group) == "gloo"


# ======================================================================================================
# MPPI
# ======================================================================================================
class HipShardBackend:
    """Per-rank MPPI compute on the HIP path (ros-turtlebot-navigation_amd/mppi.py handle)."""

    def __init__(self, mppi, device: torch.device):
        self.m = mppi
        self.device = device
        self.T, self.S = mppi.steps, mppi.records_per_step
        self.records = torch.zeros(self.T, self.S, capi.TBNAV_MPPI_REC, dtype=torch.float64, device=device)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def partials(self, x0, noise) -> torch.Tensor:
        """noise = (d_duL, d_duR) device pointers, layout [T][K_local] — or ("rng", seed, tick): the shard draws its own
        perturbations on the device (MPPI.setRngShard gives it its place in the ensemble's counter space)."""
        if noise[0] == "rng":
            self.m.shardPartialsRng(x0, int(noise[1]), int(noise[2]), self.records.data_ptr(), self._stream())
        else:
            d_l, d_r = noise
            self.m.shardPartials(x0, d_l, d_r, self.records.data_ptr(), self._stream())
        return self.records

    def combine(self, records_all: torch.Tensor, n_shards: int):
        self._keep = records_all  # keep the buffer alive until the enqueued kernel has run
        self.m.shardCombine(records_all.data_ptr(), n_shards, self._stream())

    def result(self):
        return self.m.lastControls(self._stream())


class ShardedMPPI:
    """newControls over `world_size` shards: partials -> one all-gather -> combine."""

    def __init__(self, backend, group=None):
        self.b = backend
        self.group = group
        self.world = _world(group)
        self._gathered = None
        self._gloo = _backend_is_gloo(group)  # (looked up once: a tick is a few tens of microseconds)

    def tick(self, x0, noise):
        rec = self.b.partials(x0, noise)
        if self.world == 1:
            self.b.combine(rec, 1)
            return
        if self._gloo and rec.is_cuda:
            # gloo moves host memory: stage the (tiny) record set through the CPU
            host = rec.cpu().reshape(-1)
            out = torch.empty(self.world * host.numel(), dtype=host.dtype)
            dist.all_gather_into_tensor(out, host, group=self.group)
            self.b.combine(out.to(rec.device), self.world)
            return
        flat = rec.reshape(-1)  # flat in / flat out: the layout [shard][T][S][8] both RCCL and gloo accept
        if self._gathered is None or self._gathered.numel() != self.world * flat.numel():
            self._gathered = torch.empty(self.world * flat.numel(), dtype=rec.dtype, device=rec.device)
        dist.all_gather_into_tensor(self._gathered, flat, group=self.group)
        self.b.combine(self._gathered, self.world)

    def result(self):
        return self.b.result()


# ======================================================================================================
# RBPF
# ======================================================================================================
def _all_gather_flat(out: torch.Tensor, local: torch.Tensor, group=None):
    """all_gather_into_tensor; device tensors ride RCCL as they are, and are staged through the host only when the
    process group is gloo (the CPU-side tests: two ranks sharing one GPU)."""
    if _backend_is_gloo(group) and local.is_cuda:
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h_out, local.cpu(), group=group)
        out.copy_(h_out)
    else:
        dist.all_gather_into_tensor(out, local, group=group)


class HipRbpfShardBackend:
    """Per-rank RBPF compute on the HIP path (ros-turtlebot-navigation_amd/rbpf.py handle).  Everything the exchange
    touches stays in device memory: weights, the global normalise / selection, particle blobs."""

    def __init__(self, pf, device: torch.device | None = None):
        self.pf = pf
        self.n_local = pf.N
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._L, self._h = pf._L, pf._h

    def slam_local(self, scan, u, cur_odom, prev_odom, icp_ok, T_icp, normals_local):
        return self.pf.SLAM(scan, u, cur_odom, prev_odom, icp_ok, T_icp, normals_local, local_only=True)

    def set_rng_shard(self, first_particle: int, particles_global: int):
        self.pf.setRngShard(first_particle, particles_global)

    def weights_tensor(self) -> torch.Tensor:
        w = torch.empty(self.n_local, dtype=torch.float64, device=self.device)
        capi.check(self._L.tbnav_rbpf_copy_weights_dev(self._h, w.data_ptr()), "copy_weights_dev")
        return w

    def resample(self, w_all: torch.Tensor, offset: int, z: float):
        """Sequential normalise / Neff / selection over the global vector, on the device; this rank's slice of the
        normalised weights lands in the handle.  Returns (stats, parents or None)."""
        import ctypes as C
        n = w_all.numel()
        # the handle launches on its OWN stream: a collective that produced w_all is ordered before torch's current
        # stream only (RCCL runs on the communicator's stream and does not block the host), so wait for it here
        if w_all.is_cuda:
            torch.cuda.current_stream(w_all.device).synchronize()
        parents = np.empty(n, dtype=np.int32)
        st = capi.RbpfStats()
        capi.check(self._L.tbnav_rbpf_resample_global_dev(self._h, w_all.data_ptr(), n, offset, float(z), parents.ctypes.data, C.byref(st)),
                   "resample_global_dev")
        return st, (parents if st.resampled else None)

    def set_weights_after_resample(self, global_parents_of_my_slots: np.ndarray):
        gp = np.ascontiguousarray(global_parents_of_my_slots, dtype=np.int32)
        capi.check(self._L.tbnav_rbpf_set_weights_from_global_dev(self._h, gp.ctypes.data), "set_weights_from_global_dev")

    def export_batch(self, slots):
        """The particles in `slots` (local indices; repeats allowed) as ONE device buffer, blobs back to back.
        Returns (buffer, offsets[n + 1]).  Two launches whatever n."""
        n = len(slots)
        offs = np.zeros(n + 1, dtype=np.uint64)
        if n == 0:
            return torch.empty(0, dtype=torch.uint8, device=self.device), offs.astype(np.int64)
        a = np.ascontiguousarray(slots, dtype=np.int32)
        sizes = np.zeros(n, dtype=np.uint64)
        capi.check(self._L.tbnav_rbpf_export_batch_sizes(self._h, n, a.ctypes.data, sizes.ctypes.data), "export_batch_sizes")
        total = int(sizes.sum())
        buf = torch.empty(total, dtype=torch.uint8, device=self.device)
        capi.check(self._L.tbnav_rbpf_export_batch_dev(self._h, n, a.ctypes.data, buf.data_ptr(), total, offs.ctypes.data), "export_batch_dev")
        assert int(offs[n]) == total
        return buf, offs.astype(np.int64)

    def new_blob(self, nbytes: int) -> torch.Tensor:
        return torch.empty(nbytes, dtype=torch.uint8, device=self.device)

    def import_batch(self, slots, buf: torch.Tensor, offsets):
        """Slot slots[i] takes the blob at buf + offsets[i] (several slots may name one blob)."""
        n = len(slots)
        if n == 0:
            return
        torch.cuda.synchronize(self.device)  # the receive ran on the communicator's stream, the handle has its own
        a = np.ascontiguousarray(slots, dtype=np.int32)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        capi.check(self._L.tbnav_rbpf_import_batch_dev(self._h, n, a.ctypes.data, buf.data_ptr(), buf.numel(), o.ctypes.data), "import_batch_dev")

    def gather_local(self, local_parent: np.ndarray):
        capi.check(self._L.tbnav_rbpf_gather_local(self._h, np.ascontiguousarray(local_parent, dtype=np.int32).ctypes.data),
                   "gather_local")


class ShardedRBPF:
    """ParticleFilter::SLAM over `world_size` particle shards (equal shard sizes).

    Backend protocol (HipRbpfShardBackend above; the CPU tests plug in a numpy stand-in): slam_local, weights_tensor,
    resample, set_weights_after_resample, export_batch, new_blob, import_batch, gather_local."""

    def __init__(self, backend, group=None):
        self.b = backend
        self.group = group
        self.world, self.rank = _world(group), _rank(group)
        self.n_local = backend.n_local
        self.n_global = self.n_local * self.world
        self.bytes_migrated = 0
        # production mode (device-drawn normals): every rank draws ITS slice of the ensemble's stream — without this, ranks that
        # share a seed draw identical normals and the copies of a migrated particle evolve identically (round-2 advisor finding)
        if hasattr(backend, "set_rng_shard"):
            backend.set_rng_shard(self.rank * self.n_local, self.n_global)

    def normals_slice(self, normals_global: np.ndarray, stride: int) -> np.ndarray:
        """This rank's part of the reference's draw stream (particle-major) + the resampling offset."""
        lo = self.rank * self.n_local * stride
        return np.concatenate([normals_global[lo:lo + self.n_local * stride], normals_global[-1:]])

    def tick(self, scan, u, cur_odom, prev_odom, icp_ok, T_icp, normals_global, stride):
        st_local = self.b.slam_local(scan, u, cur_odom, prev_odom, icp_ok, T_icp,
                                     None if normals_global is None else self.normals_slice(normals_global, stride))
        w_local = self.b.weights_tensor()
        if self.world > 1:
            w_all = torch.empty(self.n_global, dtype=w_local.dtype, device=w_local.device)
            _all_gather_flat(w_all, w_local, self.group)   # the ONE collective of the update (N doubles)
        else:
            w_all = w_local
        lo = self.rank * self.n_local
        # production mode: NaN = "the ensemble's offset the handle drew with its normals" (the same value on every rank)
        z = float(normals_global[-1]) if normals_global is not None else (float("nan") if hasattr(self.b, "set_rng_shard") else self.resample_offset())
        st, parents = self.b.resample(w_all, lo, z)
        if parents is None:
            return st, st_local, np.arange(self.n_global, dtype=np.int32)
        self._migrate(parents)
        # weights are NOT reset by the reference: every slot carries its parent's normalised weight
        self.b.set_weights_after_resample(parents[lo:lo + self.n_local])
        return st, st_local, parents

    def resample_offset(self) -> float:
        """Production mode (device-drawn normals): the one standard normal of lowVarianceResampling must be the same on
        every rank — rank 0 draws it."""
        t = torch.randn(1, dtype=torch.float64)
        if self.world > 1:
            if not _backend_is_gloo(self.group):
                t = t.to(self.b.device)
            dist.broadcast(t, 0, group=self.group)
        return float(t.item())

    def _migrate(self, parents: np.ndarray):
        """Slot m (global) takes the state of particle parents[m].  Parents on this rank are gathered inside the
        handle (tile tables + reference counts); parents on another rank arrive as device buffers — a particle is its
        state, counts and the tiles it owns.  ONE export of everything this rank sends, one message per destination,
        one receive buffer, one import: a handful of launches and host round trips per rank whatever the number of
        particles that move (hundreds per rank in a resample of 1000 particles per rank)."""
        nl, me = self.n_local, self.rank
        lo = me * nl
        sends = sorted({(m // nl, int(q)) for m, q in enumerate(parents) if q // nl == me and m // nl != me})   # (dst, q)
        recvs = sorted({(int(parents[m]) // nl, int(parents[m])) for m in range(lo, lo + nl) if parents[m] // nl != me})  # (src, q)
        buf, offs = self.b.export_batch([q - lo for _, q in sends])  # BEFORE anything is overwritten; blobs in (dst, q) order
        rbuf, roffs = None, np.zeros(len(recvs) + 1, dtype=np.int64)
        if self.world > 1:
            # sizes first: one small all-gather (a particle's blob depends on how many tiles it owns)
            mine = torch.zeros(nl, dtype=torch.int64)
            for i, (_, q) in enumerate(sends):
                mine[q - lo] = int(offs[i + 1] - offs[i])
            stage_cpu = _backend_is_gloo(self.group)
            sizes = torch.empty(self.n_global, dtype=torch.int64, device=mine.device if stage_cpu else self.b.device)
            _all_gather_flat(sizes, mine if stage_cpu else mine.to(self.b.device), self.group)
            sizes = sizes.cpu().numpy()
            roffs[1:] = np.cumsum([int(sizes[q]) for _, q in recvs])
            rbuf = torch.empty(int(roffs[-1]), dtype=torch.uint8) if stage_cpu else self.b.new_blob(int(roffs[-1]))
            ops, keep = [], []

            def groups(pairs):  # runs of equal peer in a list sorted by (peer, particle)
                i = 0
                while i < len(pairs):
                    j = i
                    while j < len(pairs) and pairs[j][0] == pairs[i][0]:
                        j += 1
                    yield pairs[i][0], i, j
                    i = j
            for dst, i, j in groups(sends):
                t = buf[int(offs[i]):int(offs[j])]
                t = t.cpu() if (stage_cpu and t.is_cuda) else t
                keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, dst, group=self.group))
                self.bytes_migrated += t.numel()
            for src, i, j in groups(recvs):
                ops.append(dist.P2POp(dist.irecv, rbuf[int(roffs[i]):int(roffs[j])], src, group=self.group))
            if ops:
                for r in dist.batch_isend_irecv(ops):
                    r.wait()
        # local parents first (inside the handle), then the imported ones
        local_parent = np.array([int(parents[m]) - lo if parents[m] // nl == me else -1 for m in range(lo, lo + nl)], dtype=np.int32)
        self.b.gather_local(local_parent)
        failed = None
        if recvs:
            where = {q: int(roffs[i]) for i, (_, q) in enumerate(recvs)}
            slots = [m - lo for m in range(lo, lo + nl) if parents[m] // nl != me]
            dev = getattr(self.b, "device", torch.device("cpu"))
            try:
                self.b.import_batch(slots, rbuf if rbuf.device == dev else rbuf.to(dev), [where[int(parents[lo + sl])] for sl in slots])
            except capi.TbnavError as e:
                failed = e
        # a rank whose import failed (tile pool exhausted) must not leave its peers waiting in the next collective: agree on it
        if self.world > 1:
            flag = torch.tensor([1 if failed is not None else 0], dtype=torch.int32)
            if not _backend_is_gloo(self.group):
                flag = flag.to(self.b.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            if int(flag.item()) and failed is None:
                raise capi.TbnavError(capi.ERR_POOL_EXHAUSTED, "ShardedRBPF._migrate", "another rank could not import its particles")
        if failed is not None:
            raise failed
