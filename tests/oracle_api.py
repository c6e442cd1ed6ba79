"""ctypes wrapper of oracle/liboracle.so (and oracle/_ref/libtbnav_ref.so when present).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
REF_LIB = os.path.join(ORACLE_DIR, "_ref", "libtbnav_ref.so")

_lib = None


class MppiParams(C.Structure):  # same layout as tbnav_mppi_params
    _fields_ = [
        ("wheel_radius", C.c_double), ("wheel_base", C.c_double), ("lam", C.c_double),
        ("max_wheel_vel", C.c_double), ("ul_var", C.c_double), ("ur_var", C.c_double),
        ("horizon", C.c_double), ("dt", C.c_double),
        ("Q", C.c_double * 3), ("R", C.c_double * 2), ("P1", C.c_double * 3),
        ("rollouts", C.c_int32), ("device", C.c_int32),
    ]


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".hpp"))]
        stale = (not os.path.exists(LIB)) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs)
        if stale:
            subprocess.run(["make", "liboracle.so"], cwd=ORACLE_DIR, check=True, stdout=subprocess.DEVNULL)
        _lib = C.CDLL(LIB)
        _lib.orc_loss.restype = C.c_double
        _lib.orc_terminal_loss.restype = C.c_double
        _lib.orc_mppi_steps.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def mppi_params(d: dict) -> MppiParams:
    p = MppiParams()
    for k in ("wheel_radius", "wheel_base", "lam", "max_wheel_vel", "ul_var", "ur_var", "horizon", "dt"):
        setattr(p, k, float(d[k]))
    p.Q[:] = d["Q"]; p.R[:] = d["R"]; p.P1[:] = d["P1"]
    p.rollouts = int(d["rollouts"]); p.device = -1
    return p


def normal_stream(seed: int, n: int, mu: float, sigma: float) -> np.ndarray:
    out = np.empty(n)
    lib().orc_normal_stream(C.c_uint64(seed), C.c_int64(n), C.c_double(mu), C.c_double(sigma), _p(out))
    return out


def mppi_steps(d: dict) -> int:
    p = mppi_params(d)
    return lib().orc_mppi_steps(C.byref(p))


def rk4_step(d: dict, x, u):
    p = mppi_params(d)
    xa = np.array(x, dtype=np.float64); ua = np.array(u, dtype=np.float64)
    lib().orc_rk4_step(C.byref(p), _p(xa), _p(ua))
    return xa


def loss(d, x, xd, u):
    p = mppi_params(d)
    a, b, c = (np.array(v, dtype=np.float64) for v in (x, xd, u))
    return lib().orc_loss(C.byref(p), _p(a), _p(b), _p(c))


def terminal_loss(d, x, xd):
    p = mppi_params(d)
    a, b = (np.array(v, dtype=np.float64) for v in (x, xd))
    return lib().orc_terminal_loss(C.byref(p), _p(a), _p(b))


def softmin_step(lam, Jrow, dul, dur):
    Jrow, dul, dur = (np.ascontiguousarray(v, dtype=np.float64) for v in (Jrow, dul, dur))
    K = Jrow.size
    w = np.empty(K); sl = C.c_double(); sr = C.c_double()
    lib().orc_softmin_step(C.c_double(lam), C.c_int(K), _p(Jrow), _p(dul), _p(dur), _p(w), C.byref(sl), C.byref(sr))
    return w, sl.value, sr.value


def mppi_new_controls(d: dict, u: np.ndarray, uinit, xd, x0, noise: np.ndarray) -> dict:
    """Runs one reference tick.  `u` ([2][T]) is NOT modified; the result dict carries
    loss, J (before min-subtraction), u_upd (before shift), u (after shift), out (ul, ur)."""
    p = mppi_params(d)
    T, K = lib().orc_mppi_steps(C.byref(p)), p.rollouts
    u2 = np.array(u, dtype=np.float64, order="C").copy()
    noise = np.ascontiguousarray(noise, dtype=np.float64)
    assert noise.size == K * T * 2 and u2.shape == (2, T)
    lossm = np.empty((T, K)); J = np.empty((T, K)); uupd = np.empty((2, T)); out = np.empty(2)
    ui, xdv, x0v = (np.array(v, dtype=np.float64) for v in (uinit, xd, x0))
    lib().orc_mppi_new_controls(C.byref(p), _p(u2), _p(ui), _p(xdv), _p(x0v), _p(noise), _p(lossm), _p(J),
                                _p(uupd), _p(out))
    return dict(loss=lossm, J=J, u_upd=uupd, u=u2, out=(out[0], out[1]))


def mppi_shard_partials(d: dict, u, xd, x0, noise):
    p = mppi_params(d)
    T, K = lib().orc_mppi_steps(C.byref(p)), p.rollouts
    u2 = np.ascontiguousarray(u, dtype=np.float64); noise = np.ascontiguousarray(noise, dtype=np.float64)
    J = np.empty((T, K)); rec = np.empty((T, 8))
    xdv, x0v = (np.array(v, dtype=np.float64) for v in (xd, x0))
    lib().orc_mppi_shard_partials(C.byref(p), _p(u2), _p(xdv), _p(x0v), _p(noise), _p(J), _p(rec))
    return J, rec


def mppi_combine(d: dict, u, uinit, records_all):
    """records_all: [n_rec][T][8].  Returns (u after shift, (ul, ur))."""
    p = mppi_params(d)
    u2 = np.array(u, dtype=np.float64, order="C").copy()
    rec = np.ascontiguousarray(records_all, dtype=np.float64)
    ui = np.array(uinit, dtype=np.float64); out = np.empty(2)
    lib().orc_mppi_combine(C.byref(p), _p(u2), _p(ui), _p(rec), C.c_int(rec.shape[0]), _p(out))
    return u2, (out[0], out[1])
