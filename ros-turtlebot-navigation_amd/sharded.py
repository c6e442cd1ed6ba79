"""Ensemble sharding of the MPPI tick across ranks (one process per GPU, torch.distributed).

SURVEY.md section 8-e: rollouts are independent until the per-time-step soft-min, so each rank
rolls out its K/P slice and emits T small partial records (include/tbnav_mppi.h); ONE all-gather of
those records (RCCL over xGMI on GPUs; a few KB, latency-bound) is the only collective; every rank
then runs the same combine and holds the same warm-start controls.  No data-path collective
touches the noise or cost arrays.

The per-rank compute is behind a two-method backend so the exchange logic can be exercised on CPU
with the gloo backend (tests/test_sharded_gloo.py plugs the oracle in); on GPUs the backend is the
HIP handle.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import capi


class HipShardBackend:
    """Per-rank compute on the HIP path (ros-turtlebot-navigation_amd/mppi.py handle)."""

    def __init__(self, mppi, device: torch.device):
        self.m = mppi
        self.device = device
        self.T, self.S = mppi.steps, mppi.records_per_step
        self.records = torch.zeros(self.T, self.S, capi.TBNAV_MPPI_REC, dtype=torch.float64, device=device)

    def partials(self, x0, noise) -> torch.Tensor:
        d_l, d_r = noise  # device pointers (ints), layout [T][K_local]
        st = torch.cuda.current_stream(self.device).cuda_stream
        self.m.shardPartials(x0, d_l, d_r, self.records.data_ptr(), st)
        return self.records

    def combine(self, records_all: torch.Tensor, n_shards: int):
        st = torch.cuda.current_stream(self.device).cuda_stream
        self.m.shardCombine(records_all.data_ptr(), n_shards, st)

    def result(self):
        return self.m.lastControls(torch.cuda.current_stream(self.device).cuda_stream)


class ShardedMPPI:
    """newControls over `world_size` shards: partials -> one all-gather -> combine."""

    def __init__(self, backend, group=None):
        self.b = backend
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._gathered = None

    def tick(self, x0, noise):
        rec = self.b.partials(x0, noise)
        if self.world == 1:
            self.b.combine(rec, 1)
            return
        if self._gathered is None or self._gathered.shape[1:] != rec.shape:
            self._gathered = torch.empty((self.world,) + tuple(rec.shape), dtype=rec.dtype, device=rec.device)
        dist.all_gather_into_tensor(self._gathered, rec, group=self.group)
        self.b.combine(self._gathered, self.world)

    def result(self):
        return self.b.result()
