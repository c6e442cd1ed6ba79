"""Python mirror of controller::MPPI over the C-ABI (tests / smoke / bench plumbing).

Same names and argument meaning as the reference's class surface
(controller/include/controller/mppi.hpp:31-185): CartModel, LossFunc, MPPI with
setInitialControls / setWaypoint / newControls.  The shipped host surface for ROS nodes is the C++
shim in host/; this module exists so the parity tests read like the reference's own call sites
(nuturtle_robot/src/mppi_waypoints_node.cpp:186-199,216,257,265).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import capi


@dataclass
class CartModel:  # mppi.hpp:31-52
    wheel_radius: float
    wheel_base: float


@dataclass
class LossFunc:  # mppi.hpp:56-110 (diagonals only, like the reference's ctor)
    Q: list = field(default_factory=lambda: [0.0, 0.0, 0.0])
    R: list = field(default_factory=lambda: [0.0, 0.0])
    P1: list = field(default_factory=lambda: [0.0, 0.0, 0.0])

    def __post_init__(self):
        # std::vector::at throws std::out_of_range on short vectors (mppi.hpp:68-79)
        if len(self.Q) < 3 or len(self.R) < 2 or len(self.P1) < 3:
            raise IndexError("LossFunc: Q, P1 need 3 entries and R needs 2")


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class MPPI:
    """controller::MPPI (mppi.hpp:119-185) on one MI355X."""

    def __init__(self, cart_model: CartModel, loss_func: LossFunc, lam: float, max_wheel_vel: float,
                 ul_var: float, ur_var: float, horizon: float, dt: float, rollouts: int,
                 device: int = -1, keep_j: bool = True, kernel: int | str | None = None):
        """keep_j: the fused small-K kernel also stores the cost-to-go so that costToGo() works (test plumbing; the
        bench and the C++ class leave it off).  kernel: TBNAV_MPPI_OPT_KERNEL value, or "scan" for the time-parallel
        three-kernel tick with its automatic chunk size."""
        p = capi.MppiParams()
        p.wheel_radius, p.wheel_base = cart_model.wheel_radius, cart_model.wheel_base
        p.lam, p.max_wheel_vel, p.ul_var, p.ur_var = lam, max_wheel_vel, ul_var, ur_var
        p.horizon, p.dt = horizon, dt
        p.Q[:] = loss_func.Q[:3]
        p.R[:] = loss_func.R[:2]
        p.P1[:] = loss_func.P1[:3]
        p.rollouts, p.device = rollouts, device
        self.params = p
        self._L = capi.lib()
        self._h = C.c_void_p()
        capi.check(self._L.tbnav_mppi_create(C.byref(p), C.byref(self._h)), "tbnav_mppi_create")
        self.steps = self._L.tbnav_mppi_steps(self._h)
        if kernel == "scan":
            kernel = next(tc for tc in (4, 5, 6, 8, 10, 12, 16, 20) if -(-self.steps // tc) <= 12)
        if kernel is not None:
            self.setOption(capi.MPPI_OPT_KERNEL, int(kernel))
        if keep_j:
            self.setOption(capi.MPPI_OPT_KEEP_J, 1)
        self.rollouts = self._L.tbnav_mppi_rollouts(self._h)
        self.records_per_step = self._L.tbnav_mppi_records_per_step(self._h)
        self._name_kernel()

    def _name_kernel(self):
        v = self._L.tbnav_mppi_rollout_variant(self._h)
        seq = ("mppi_rollout_cost", "mppi_rollout_cost", "mppi_rollout_prefix")[max(0, self._L.tbnav_mppi_streaming_form(self._h))]
        self.rollout_kernel = (seq if v == 0 else f"mppi_rollout_scan<{v} steps/thread>" if v > 0
                               else f"mppi_rollout_fused<{-v} rollouts/workgroup> (rollout + partial records)")

    def setOption(self, option: int, value: int):
        capi.check(self._L.tbnav_mppi_set_option(self._h, option, value), "tbnav_mppi_set_option")
        self._name_kernel()

    def setRngShard(self, first_rollout: int, rollouts_global: int):
        capi.check(self._L.tbnav_mppi_set_rng_shard(self._h, first_rollout, rollouts_global), "tbnav_mppi_set_rng_shard")

    def graphReplayedTicks(self) -> int:
        return int(self._L.tbnav_mppi_graph_replayed_ticks(self._h))

    def attachComm(self, comm):
        """Every tick of this handle becomes the sharded tick (shard partials -> one RCCL all-gather of the records -> combine
        of all shards) inside the library, on the tick's stream (rtn_amd.comm.Comm; None detaches)."""
        capi.check(self._L.tbnav_mppi_attach_comm(self._h, comm._h if comm is not None else None), "tbnav_mppi_attach_comm")
        self._comm = comm  # (the communicator must outlive the handle's ticks)

    def exchangeKind(self) -> int:
        """0: no communicator; 1: the communicator's all-gather; 2: direct stores into the peers' buffers (tbnav_mppi_exchange_kind)."""
        return int(self._L.tbnav_mppi_exchange_kind(self._h))

    def exchangeProbe(self, rounds: int, stream: int = 0) -> float:
        """Microseconds per exchange of the record block alone (collective over the communicator's ranks; tbnav_mppi_exchange_probe)."""
        us = C.c_double()
        capi.check(self._L.tbnav_mppi_exchange_probe(self._h, rounds, stream or None, C.byref(us)), "exchange_probe")
        return us.value

    def setDirectExchange(self, on: bool):
        """TBNAV_MPPI_OPT_DIRECT_EXCHANGE; before attachComm."""
        capi.check(self._L.tbnav_mppi_set_option(self._h, 8, int(on)), "set_option(direct exchange)")  # (2: fault injection, tests)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value and getattr(self, "_owned", True):
            self._L.tbnav_mppi_destroy(self._h)
        self._h = C.c_void_p()

    __del__ = close

    # ---- reference surface ----
    def setInitialControls(self, uL: float, uR: float):
        capi.check(self._L.tbnav_mppi_set_initial_controls(self._h, uL, uR), "set_initial_controls")

    def setDynamics(self, model: str):
        """"rk4": the reference's CartModel + RK4 (default).  "arc": every rollout step is the plant's own update,
        DiffDrive::feedforward(wheelsToTwist(u) * dt) — exact arcs (an option, not the reference)."""
        capi.check(self._L.tbnav_mppi_set_dynamics(self._h, {"rk4": 0, "arc": 1}[model]), "set_dynamics")

    def setWaypoint(self, x: float, y: float, theta: float):
        capi.check(self._L.tbnav_mppi_set_waypoint(self._h, x, y, theta), "set_waypoint")

    def newControls(self, x: float, y: float, theta: float, noise: np.ndarray):
        """noise: host array [K][T][2] in the reference's draw order.  Returns (ul, ur)."""
        noise = np.ascontiguousarray(noise, dtype=np.float64)
        assert noise.size == self.rollouts * self.steps * 2
        x0 = (C.c_double * 3)(x, y, theta)
        out = (C.c_double * 2)()
        capi.check(self._L.tbnav_mppi_new_controls(self._h, x0, noise.ctypes.data, out), "new_controls")
        return out[0], out[1]

    # ---- device-resident variants ----
    def newControlsDev(self, x0, d_duL: int, d_duR: int, stream: int = 0):
        x0c = (C.c_double * 3)(*x0)
        out = (C.c_double * 2)()
        capi.check(self._L.tbnav_mppi_new_controls_dev(self._h, x0c, d_duL or None, d_duR or None,
                                                       stream or None, out), "new_controls_dev")
        return out[0], out[1]

    def enqueueDev(self, x0, d_duL: int, d_duR: int, stream: int = 0):
        x0c = (C.c_double * 3)(*x0)
        capi.check(self._L.tbnav_mppi_enqueue_dev(self._h, x0c, d_duL or None, d_duR or None,
                                                  stream or None), "enqueue_dev")

    def profileTick(self, x0, d_duL: int, d_duR: int, stream: int = 0):
        """(ms_rollout_cost, ms_partials, ms_combine) of one tick, HIP events on `stream`."""
        x0c = (C.c_double * 3)(*x0)
        ms = (C.c_float * 3)()
        capi.check(self._L.tbnav_mppi_profile_tick(self._h, x0c, d_duL or None, d_duR or None,
                                                   stream or None, ms), "profile_tick")
        return ms[0], ms[1], ms[2]

    def profileKernels(self, x0, d_duL: int, d_duR: int, stream: int = 0, reps: int = 100):
        """(ms_rollout, ms_partials, ms_combine): each kernel launched `reps` times back to back between one event pair."""
        x0c = (C.c_double * 3)(*x0)
        ms = (C.c_float * 3)()
        capi.check(self._L.tbnav_mppi_profile_kernels(self._h, x0c, d_duL or None, d_duR or None, stream or None, reps, ms),
                   "profile_kernels")
        return ms[0], ms[1], ms[2]

    def profileKernelsRng(self, x0, seed: int, tick: int, stream: int = 0, reps: int = 100):
        """The same for the production tick: the fused kernel's in-kernel-noise instantiation where that is what the tick runs."""
        ms = (C.c_float * 3)()
        capi.check(self._L.tbnav_mppi_profile_kernels_rng(self._h, (C.c_double * 3)(*x0), seed, tick, stream or None, reps, ms),
                   "profile_kernels_rng")
        return ms[0], ms[1], ms[2]

    def lastKernelNames(self):
        """(rollout, combine): the instantiations the last launches were, as rocprofv3 prints them."""
        a, b = C.create_string_buffer(96), C.create_string_buffer(96)
        capi.check(self._L.tbnav_mppi_last_kernel_names(self._h, a, 96, b, 96), "last_kernel_names")
        return a.value.decode(), b.value.decode()

    def lastControls(self, stream: int = 0):
        out = (C.c_double * 2)()
        capi.check(self._L.tbnav_mppi_last_controls(self._h, stream or None, out), "last_controls")
        return out[0], out[1]

    def enqueueRng(self, x0, seed: int, tick: int, stream: int = 0):
        """Production tick: perturbations of (seed, tick) drawn on the device, inside the fused kernel where there is one."""
        capi.check(self._L.tbnav_mppi_enqueue_rng(self._h, (C.c_double * 3)(*x0), seed, tick, stream or None), "enqueue_rng")

    def enqueueRngBatch(self, x0, seed: int, first_tick: int, n_ticks: int, stream: int = 0):
        """n_ticks production ticks in a row from the same state x0, enqueued from C (tbnav_mppi_enqueue_rng_batch)."""
        capi.check(self._L.tbnav_mppi_enqueue_rng_batch(self._h, (C.c_double * 3)(*x0), 0, seed, first_tick, n_ticks, stream or None),
                   "enqueue_rng_batch")

    def newControlsRng(self, x0, seed: int, tick: int, stream: int = 0):
        out = (C.c_double * 2)()
        capi.check(self._L.tbnav_mppi_new_controls_rng(self._h, (C.c_double * 3)(*x0), seed, tick, stream or None, out),
                   "new_controls_rng")
        return out[0], out[1]

    def sampleNoise(self, seed: int, tick: int, stream: int = 0):
        capi.check(self._L.tbnav_mppi_sample_noise(self._h, seed, tick, stream or None), "sample_noise")

    def getNoise(self):
        a = np.empty((self.steps, self.rollouts)); b = np.empty_like(a)
        capi.check(self._L.tbnav_mppi_get_noise(self._h, a.ctypes.data, b.ctypes.data), "get_noise")
        return a, b

    def shardPartials(self, x0, d_duL: int, d_duR: int, d_records: int, stream: int = 0):
        x0c = (C.c_double * 3)(*x0)
        capi.check(self._L.tbnav_mppi_shard_partials(self._h, x0c, d_duL or None, d_duR or None,
                                                     stream or None, d_records), "shard_partials")

    def shardPartialsRng(self, x0, seed: int, tick: int, d_records: int, stream: int = 0):
        x0c = (C.c_double * 3)(*x0)
        capi.check(self._L.tbnav_mppi_shard_partials_rng(self._h, x0c, seed, tick, stream or None, d_records), "shard_partials_rng")

    def shardCombine(self, d_records_all: int, n_shards: int, stream: int = 0):
        capi.check(self._L.tbnav_mppi_shard_combine(self._h, d_records_all, n_shards, stream or None),
                   "shard_combine")

    # ---- state / debug ----
    def getControls(self) -> np.ndarray:
        u = np.empty((2, self.steps))
        capi.check(self._L.tbnav_mppi_get_controls(self._h, u.ctypes.data), "get_controls")
        return u

    def setControls(self, u: np.ndarray):
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert u.shape == (2, self.steps)
        capi.check(self._L.tbnav_mppi_set_controls(self._h, u.ctypes.data), "set_controls")

    def costToGo(self) -> np.ndarray:
        J = np.empty((self.steps, self.rollouts))
        capi.check(self._L.tbnav_mppi_get_cost_to_go(self._h, J.ctypes.data), "get_cost_to_go")
        return J


class MPPIGroup:
    """tbnav_mppi_group: ONE process driving the ensemble over several devices (what controller::MPPI(..., n_gpus) holds).
    rollouts is the ensemble's K; devices may repeat (members sharing a device exchange by copies instead of RCCL)."""

    def __init__(self, cart_model: CartModel, loss_func: LossFunc, lam: float, max_wheel_vel: float, ul_var: float, ur_var: float,
                 horizon: float, dt: float, rollouts: int, devices, keep_j: bool = True):
        p = capi.MppiParams()
        p.wheel_radius, p.wheel_base = cart_model.wheel_radius, cart_model.wheel_base
        p.lam, p.max_wheel_vel, p.ul_var, p.ur_var = lam, max_wheel_vel, ul_var, ur_var
        p.horizon, p.dt = horizon, dt
        p.Q[:] = loss_func.Q[:3]; p.R[:] = loss_func.R[:2]; p.P1[:] = loss_func.P1[:3]
        p.rollouts, p.device = rollouts, -1
        self._L = capi.lib()
        self._h = C.c_void_p()
        d = np.ascontiguousarray(devices, dtype=np.int32)
        capi.check(self._L.tbnav_mppi_group_create(C.byref(p), len(d), d.ctypes.data, C.byref(self._h)), "tbnav_mppi_group_create")
        self.n = len(d)
        self.rollouts = rollouts
        self.steps = self._L.tbnav_mppi_steps(self._member_handle(0))
        if keep_j:
            self.setOption(capi.MPPI_OPT_KEEP_J, 1)

    def _member_handle(self, r: int) -> C.c_void_p:
        h = C.c_void_p()
        capi.check(self._L.tbnav_mppi_group_member(self._h, r, C.byref(h)), "tbnav_mppi_group_member")
        return h

    def member(self, r: int) -> MPPI:
        """A borrowed view of shard r (parity hooks: costToGo, getControls ...)."""
        m = MPPI.__new__(MPPI)
        m._L, m._h, m._owned = self._L, self._member_handle(r), False
        m.steps = self.steps
        m.rollouts = self._L.tbnav_mppi_rollouts(m._h)
        m.records_per_step = self._L.tbnav_mppi_records_per_step(m._h)
        m._name_kernel()
        return m

    def setOption(self, option: int, value: int):
        capi.check(self._L.tbnav_mppi_group_set_option(self._h, option, value), "tbnav_mppi_group_set_option")

    def setWaypoint(self, x, y, theta):
        capi.check(self._L.tbnav_mppi_group_set_waypoint(self._h, x, y, theta), "group_set_waypoint")

    def setInitialControls(self, uL, uR):
        capi.check(self._L.tbnav_mppi_group_set_initial_controls(self._h, uL, uR), "group_set_initial_controls")

    def setControls(self, u: np.ndarray):
        u = np.ascontiguousarray(u, dtype=np.float64)
        capi.check(self._L.tbnav_mppi_group_set_controls(self._h, u.ctypes.data), "group_set_controls")

    def getControls(self) -> np.ndarray:
        u = np.empty((2, self.steps))
        capi.check(self._L.tbnav_mppi_group_get_controls(self._h, u.ctypes.data), "group_get_controls")
        return u

    def newControls(self, x, y, theta, noise: np.ndarray):
        """noise: the ENSEMBLE's perturbations [K][T][2] in the reference's draw order."""
        noise = np.ascontiguousarray(noise, dtype=np.float64)
        assert noise.shape == (self.rollouts, self.steps, 2)
        x0 = (C.c_double * 3)(x, y, theta); out = (C.c_double * 2)()
        capi.check(self._L.tbnav_mppi_group_new_controls(self._h, x0, noise.ctypes.data, out), "group_new_controls")
        return (out[0], out[1])

    def newControlsRng(self, x0, seed: int, tick: int):
        x0c = (C.c_double * 3)(*x0); out = (C.c_double * 2)()
        capi.check(self._L.tbnav_mppi_group_new_controls_rng(self._h, x0c, seed, tick, out), "group_new_controls_rng")
        return (out[0], out[1])

    def enqueueRngBatch(self, x0, seed: int, first_tick: int, n_ticks: int):
        x0c = (C.c_double * 3)(*x0)
        capi.check(self._L.tbnav_mppi_group_enqueue_rng_batch(self._h, C.cast(x0c, C.c_void_p), 0, seed, first_tick, n_ticks), "group_enqueue_rng_batch")

    def lastControls(self):
        out = (C.c_double * 2)()
        capi.check(self._L.tbnav_mppi_group_last_controls(self._h, out), "group_last_controls")
        return (out[0], out[1])

    def synchronize(self):
        capi.check(self._L.tbnav_mppi_group_synchronize(self._h), "group_synchronize")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.tbnav_mppi_group_destroy(self._h)
        self._h = C.c_void_p()

    __del__ = close
