"""ctypes face of include/tbnav_comm.h: the communicator the sharded MPPI tick / RBPF scan exchange through INSIDE
libtbnav_hip.so (RCCL over xGMI; in-process copies for ranks of a one-process group that share a device).

Plumbing for tests and bench.py.  `Comm.from_torch_distributed()` builds this process's rank of a torch.distributed job:
rank 0 draws the RCCL unique id and the job's own process group carries its 128 bytes to the other ranks."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

ID_BYTES = 128


class Comm:
    def __init__(self, handle: C.c_void_p, owns: bool = True):
        self._L = capi.lib()
        self._h = handle
        self._owns = owns

    @staticmethod
    def unique_id(transport: str = "rccl") -> bytes:
        """transport "rccl": ncclGetUniqueId; "ipc": an id of the IPC transport (ranks = processes of one node that may share a
        device; tbnav_comm_unique_id_ipc) — tbnav_comm_create tells them apart."""
        buf = (C.c_uint8 * ID_BYTES)()
        if transport == "ipc":
            capi.check(capi.lib().tbnav_comm_unique_id_ipc(C.cast(buf, C.c_void_p)), "tbnav_comm_unique_id_ipc")
        else:
            assert transport == "rccl", transport
            capi.check(capi.lib().tbnav_comm_unique_id(C.cast(buf, C.c_void_p)), "tbnav_comm_unique_id")
        return bytes(buf)

    @classmethod
    def create(cls, uid: bytes, nranks: int, rank: int, device: int = -1) -> "Comm":
        assert len(uid) == ID_BYTES
        buf = (C.c_uint8 * ID_BYTES).from_buffer_copy(uid)
        h = C.c_void_p()
        capi.check(capi.lib().tbnav_comm_create(C.cast(buf, C.c_void_p), nranks, rank, device, C.byref(h)), "tbnav_comm_create")
        return cls(h)

    @classmethod
    def create_local(cls, devices) -> list["Comm"]:
        n = len(devices)
        d = np.ascontiguousarray(devices, dtype=np.int32)
        out = (C.c_void_p * n)()
        capi.check(capi.lib().tbnav_comm_create_local(n, d.ctypes.data, C.cast(out, C.c_void_p)), "tbnav_comm_create_local")
        return [cls(C.c_void_p(out[i])) for i in range(n)]

    @classmethod
    def from_torch_distributed(cls, device: int, group=None, transport: str = "rccl") -> "Comm":
        """This process's rank of the initialised torch.distributed job (any backend: the id travels as a byte tensor)."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        # the id and one more byte: 1 = rank 0 has an id.  A rank 0 that cannot draw one (librccl missing) still takes part in
        # the broadcast, so that EVERY rank raises here instead of rank 0 alone leaving its peers inside the collective
        t = torch.zeros(ID_BYTES + 1, dtype=torch.uint8)
        err = None
        if rank == 0:
            try:
                t[:ID_BYTES] = torch.frombuffer(bytearray(cls.unique_id(transport)), dtype=torch.uint8)
                t[ID_BYTES] = 1
            except Exception as e:  # noqa: BLE001
                err = e
        if dist.get_backend(group) != "gloo":
            t = t.to(torch.device("cuda", device))
        dist.broadcast(t, 0, group=group)
        raw = bytes(t.cpu().numpy().tobytes())
        if raw[ID_BYTES] != 1:
            raise RuntimeError(f"rank 0 could not draw a communicator id ({err})" if rank == 0 else "rank 0 could not draw a communicator id")
        return cls.create(raw[:ID_BYTES], world, rank, device)

    rank = property(lambda self: self._L.tbnav_comm_rank(self._h))
    size = property(lambda self: self._L.tbnav_comm_size(self._h))
    device = property(lambda self: self._L.tbnav_comm_device(self._h))
    uses_rccl = property(lambda self: bool(self._L.tbnav_comm_uses_rccl(self._h)))

    def selftest(self, nbytes: int = 1 << 20):
        """One all-gather and one ring of point-to-point messages through this communicator's transport, checked (collective)."""
        capi.check(self._L.tbnav_comm_selftest(self._h, nbytes), "tbnav_comm_selftest")

    def close(self):
        if self._owns and getattr(self, "_h", None) is not None and self._h.value:
            self._L.tbnav_comm_destroy(self._h)
        self._h = C.c_void_p()

    __del__ = close
