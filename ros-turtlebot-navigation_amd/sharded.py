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
        d_l, d_r = noise  # device pointers (ints), layout [T][K_local]
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

    def tick(self, x0, noise):
        rec = self.b.partials(x0, noise)
        if self.world == 1:
            self.b.combine(rec, 1)
            return
        if _backend_is_gloo(self.group) and rec.is_cuda:
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
class HipRbpfShardBackend:
    """Per-rank RBPF compute on the HIP path (ros-turtlebot-navigation_amd/rbpf.py handle)."""

    def __init__(self, pf):
        self.pf = pf
        self.n_local = pf.N

    def slam_local(self, scan, u, cur_odom, prev_odom, icp_ok, T_icp, normals_local):
        return self.pf.SLAM(scan, u, cur_odom, prev_odom, icp_ok, T_icp, normals_local, local_only=True)

    def weights(self) -> np.ndarray:
        return self.pf.particles()[2]

    def set_weights(self, w: np.ndarray):
        self.pf.setParticles(w=w)

    def export_particle(self, slot: int) -> dict:
        pose, prev, w = self.pf.particles()
        return dict(state=np.concatenate([pose[slot], prev[slot], [w[slot]]]), log_odds=self.pf.logOdds(slot),
                    dist=self.pf.occDist(slot))

    def import_particle(self, slot: int, blob: dict):
        pose, prev, w = self.pf.particles()
        pose[slot], prev[slot], w[slot] = blob["state"][0:3], blob["state"][3:6], blob["state"][6]
        self.pf.setParticles(pose, prev, w)
        self.pf.setLogOdds(slot, blob["log_odds"])
        self.pf.setOccDist(slot, blob["dist"])

    def gather_local(self, local_parent: np.ndarray):
        capi.check(self.pf._L.tbnav_rbpf_gather_local(self.pf._h, np.ascontiguousarray(local_parent, dtype=np.int32).ctypes.data),
                   "gather_local")


class ShardedRBPF:
    """ParticleFilter::SLAM over `world_size` particle shards (equal shard sizes)."""

    def __init__(self, backend, resample_global, group=None):
        self.b = backend
        self.resample_global = resample_global  # callable(weights_all, z) -> (parents, w_norm, stats)
        self.group = group
        self.world, self.rank = _world(group), _rank(group)
        self.n_local = backend.n_local
        self.n_global = self.n_local * self.world

    def normals_slice(self, normals_global: np.ndarray, stride: int) -> np.ndarray:
        """This rank's part of the reference's draw stream (particle-major) + the resampling offset."""
        lo = self.rank * self.n_local * stride
        return np.concatenate([normals_global[lo:lo + self.n_local * stride], normals_global[-1:]])

    def tick(self, scan, u, cur_odom, prev_odom, icp_ok, T_icp, normals_global, stride):
        st_local = self.b.slam_local(scan, u, cur_odom, prev_odom, icp_ok, T_icp, self.normals_slice(normals_global, stride))
        w_local = torch.from_numpy(np.ascontiguousarray(self.b.weights(), dtype=np.float64))
        if self.world > 1:
            w_all = torch.empty(self.n_global, dtype=torch.float64)
            if dist.get_backend(self.group) == "nccl":
                dev = torch.device("cuda", torch.cuda.current_device())
                g = torch.empty(self.n_global, dtype=torch.float64, device=dev)
                dist.all_gather_into_tensor(g, w_local.to(dev), group=self.group)
                w_all = g.cpu()
            else:
                dist.all_gather_into_tensor(w_all, w_local, group=self.group)
        else:
            w_all = w_local
        parents, w_norm, st = self.resample_global(w_all.numpy(), float(normals_global[-1]))
        lo = self.rank * self.n_local
        if st.resampled:
            self._migrate(parents)
            # weights are NOT reset by the reference: every slot carries its parent's normalised weight
            self.b.set_weights(w_norm[parents[lo:lo + self.n_local]])
        else:
            self.b.set_weights(w_norm[lo:lo + self.n_local])
        return st, st_local, parents

    def _migrate(self, parents: np.ndarray):
        """Slot m (global) takes the state of particle parents[m].  Remote parents are exported by their
        owner and sent point-to-point; local ones are gathered inside the handle."""
        nl, me = self.n_local, self.rank
        lo = me * nl
        # what I must send: for every remote slot whose parent I own (dedup per (dst, parent))
        sends = {}
        for m, q in enumerate(parents):
            dst, src = m // nl, q // nl
            if src == me and dst != me:
                sends.setdefault((dst, int(q)), None)
        recvs = sorted({(int(parents[m]) // nl, int(parents[m])) for m in range(lo, lo + nl) if parents[m] // nl != me})
        blobs = {key: self.b.export_particle(key[1] - lo) for key in sends}  # export BEFORE anything is overwritten
        received = {}
        if self.world > 1:
            reqs = []
            for (dst, q), blob in sorted(blobs.items()):
                for name in ("state", "log_odds", "dist"):
                    reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(blob[name])), dst, group=self.group))
            sizes = None
            for (src, q) in recvs:
                if sizes is None:
                    probe = self.b.export_particle(0)
                    sizes = {k: v.size for k, v in probe.items()}
                got = {}
                for name in ("state", "log_odds", "dist"):
                    t = torch.empty(sizes[name], dtype=torch.float64)
                    dist.recv(t, src, group=self.group)
                    got[name] = t.numpy()
                received[q] = got
            for r in reqs:
                r.wait()
        # local parents first (double-buffered gather inside the handle), then the imported ones
        local_parent = np.array([int(parents[m]) - lo if parents[m] // nl == me else -1 for m in range(lo, lo + nl)], dtype=np.int32)
        self.b.gather_local(local_parent)
        for m in range(lo, lo + nl):
            q = int(parents[m])
            if q // nl != me:
                self.b.import_particle(m - lo, received[q])
