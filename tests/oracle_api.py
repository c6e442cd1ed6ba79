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


def mppi_new_controls(d: dict, u: np.ndarray, uinit, xd, x0, noise: np.ndarray, dyn: int = 0) -> dict:
    """Runs one reference tick.  `u` ([2][T]) is NOT modified; the result dict carries
    loss, J (before min-subtraction), u_upd (before shift), u (after shift), out (ul, ur).
    dyn = 1: exact-arc plant dynamics (DiffDrive::feedforward per step) instead of the RK4 cart."""
    p = mppi_params(d)
    T, K = lib().orc_mppi_steps(C.byref(p)), p.rollouts
    u2 = np.array(u, dtype=np.float64, order="C").copy()
    noise = np.ascontiguousarray(noise, dtype=np.float64)
    assert noise.size == K * T * 2 and u2.shape == (2, T)
    lossm = np.empty((T, K)); J = np.empty((T, K)); uupd = np.empty((2, T)); out = np.empty(2)
    ui, xdv, x0v = (np.array(v, dtype=np.float64) for v in (uinit, xd, x0))
    lib().orc_mppi_new_controls_dyn(C.byref(p), _p(u2), _p(ui), _p(xdv), _p(x0v), _p(noise), _p(lossm), _p(J),
                                    _p(uupd), _p(out), int(dyn))
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


# =====================================================================================================
# RBPF side: the restated grid/filter (prefix "orc_") and the REAL reference classes (prefix "ref_",
# oracle/_ref/libtbnav_ref.so — present in the build container, travels to the GPU box as a built
# file, absent otherwise).
# =====================================================================================================
_ref = None


def ref_available() -> bool:
    return os.path.exists(REF_LIB)


def ref_lib():
    global _ref
    if _ref is None:
        if not ref_available():
            raise FileNotFoundError(REF_LIB)
        _ref = C.CDLL(REF_LIB)
    return _ref


def _setup_grid_sigs(L, pre):
    dbl = C.c_double
    getattr(L, pre + "gm_create").restype = C.c_void_p
    getattr(L, pre + "gm_clone").restype = C.c_void_p
    getattr(L, pre + "gm_clone").argtypes = [C.c_void_p]
    getattr(L, pre + "gm_likelihood").restype = dbl
    getattr(L, pre + "gm_world2rowmajor").restype = C.c_int64
    getattr(L, pre + "gm_world2rowmajor").argtypes = [C.c_void_p, dbl, dbl]
    getattr(L, pre + "normalize_angle_PI").restype = dbl
    getattr(L, pre + "normalize_angle_PI").argtypes = [dbl]
    getattr(L, pre + "pdf_normal").restype = dbl
    getattr(L, pre + "pdf_normal").argtypes = [dbl, dbl, C.POINTER(C.c_int)]
    getattr(L, pre + "log_odds_to_prob").restype = dbl
    getattr(L, pre + "log_odds_to_prob").argtypes = [dbl]
    getattr(L, pre + "prob_to_log_odds").restype = dbl
    getattr(L, pre + "prob_to_log_odds").argtypes = [dbl]
    getattr(L, pre + "dd_create").restype = C.c_void_p
    getattr(L, pre + "dd_create").argtypes = [C.c_void_p, dbl, dbl]
    getattr(L, pre + "dd_update_odometry").argtypes = [C.c_void_p, dbl, dbl, C.c_void_p]


# the shipped laser / mixture / clamp parameters (bmapping/launch/slam.launch:19-42,
# bmapping/config/LDS_01_lidar.yaml; angles converted as turtle_mapping_node.cpp:300-302 does)
def lds01_laser(beam_delta_deg=1.0):
    d2r = np.pi / 180.0
    return np.array([0.0 * d2r, 360.0 * d2r, beam_delta_deg * d2r, 0.12, 3.5], dtype=np.float32)


MIX = np.array([0.95, 0.0, 0.04, 0.01, 0.5])  # z_hit, z_short, z_max, z_rand, sigma_hit


class GridAPI:
    """bmapping::GridMapper through either library.  pose arguments are (theta, x, y)."""

    def __init__(self, which="orc", grid=(0.05, -2.0, 2.0, -2.0, 2.0), laser=None, mix=MIX, trs=(0.0, 0.0, 0.0),
                 _handle=None, _borrowed=False):
        self.which = which
        self.L = lib() if which == "orc" else ref_lib()
        self.pre = which + "_"
        _setup_grid_sigs(self.L, self.pre)
        self._borrowed = _borrowed
        if _handle is not None:
            self.h = C.c_void_p(_handle)
        else:
            g = np.array(grid, dtype=np.float64)
            la = lds01_laser() if laser is None else np.asarray(laser, dtype=np.float32)
            mx = np.asarray(mix, dtype=np.float64); t = np.asarray(trs, dtype=np.float64)
            self.h = C.c_void_p(self._f("gm_create")(_p(g), _p(la), _p(mx), _p(t)))
        xs, ys = C.c_int(), C.c_int()
        self._f("gm_size")(self.h, C.byref(xs), C.byref(ys))
        self.xsize, self.ysize = xs.value, ys.value
        self.G = self.xsize * self.ysize

    def _f(self, name):
        return getattr(self.L, self.pre + name)

    def clone(self):
        return GridAPI(self.which, _handle=self._f("gm_clone")(self.h))

    def close(self):
        if self.h and not self._borrowed:
            self._f("gm_destroy")(self.h)
        self.h = None

    def constants(self):
        out = np.empty(4); self._f("gm_constants")(self.h, _p(out)); return out

    def integrate_scan(self, scan, pose, esdf=True):
        scan = np.ascontiguousarray(scan, dtype=np.float32); ps = np.array(pose, dtype=np.float64)
        fn = "gm_integrate_scan" if esdf else "gm_integrate_scan_no_esdf"
        return self._f(fn)(self.h, _p(scan), C.c_int(scan.size), _p(ps))

    def likelihood(self, scan, pose):
        scan = np.ascontiguousarray(scan, dtype=np.float32); ps = np.array(pose, dtype=np.float64)
        err = C.c_int()
        v = self._f("gm_likelihood")(self.h, _p(scan), C.c_int(scan.size), _p(ps), C.byref(err))
        return v, err.value

    def dump(self):
        lo = np.empty(self.G); pr = np.empty(self.G); od = np.empty(self.G); stt = np.empty(self.G, dtype=np.int32)
        self._f("gm_dump")(self.h, _p(lo), _p(pr), _p(od), _p(stt))
        return dict(log_odds=lo, prob=pr, occ_dist=od, state=stt)

    def set_occ_dist(self, od):
        od = np.ascontiguousarray(od, dtype=np.float64)
        self._f("gm_set_occ_dist")(self.h, _p(od))

    def occ_cells(self):
        out = np.empty(self.G, dtype=np.int32)
        n = self._f("gm_occ_cells")(self.h, _p(out), C.c_int(self.G))
        return out[:n].copy()

    def grid_map(self):
        out = np.empty(self.G, dtype=np.int8); self._f("gm_grid_map")(self.h, _p(out)); return out

    def end_points(self, scan, pose):
        scan = np.ascontiguousarray(scan, dtype=np.float32); ps = np.array(pose, dtype=np.float64)
        xy = np.empty((scan.size, 2))
        n = self._f("gm_end_points")(self.h, _p(scan), C.c_int(scan.size), _p(ps), _p(xy))
        return xy[:n].copy()

    def world2rowmajor(self, x, y):
        return int(self._f("gm_world2rowmajor")(self.h, float(x), float(y)))

    def free_index(self, point, pose):
        pt = np.array(point, dtype=np.float64); ps = np.array(pose, dtype=np.float64)
        out = np.empty(4096, dtype=np.int32)
        n = self._f("gm_free_index")(self.h, _p(pt), _p(ps), _p(out), C.c_int(out.size))
        return None if n < 0 else out[:n].copy()

    def line_cells(self, which, x0, y0, x1, y1):
        out = np.empty(4096, dtype=np.int32)
        n = self._f("gm_line_cells")(self.h, C.c_int(which), C.c_int(x0), C.c_int(y0), C.c_int(x1), C.c_int(y1), _p(out), C.c_int(out.size))
        return out[:n].copy()


class RigidAPI:
    """rigid2d free functions / Transform2D / DiffDrive through either library."""

    def __init__(self, which="orc"):
        self.L = lib() if which == "orc" else ref_lib()
        self.pre = which + "_"
        _setup_grid_sigs(self.L, self.pre)

    def _f(self, name):
        return getattr(self.L, self.pre + name)

    def normalize_angle_PI(self, r):
        return self._f("normalize_angle_PI")(float(r))

    def pdf_normal(self, a, b):
        e = C.c_int(); v = self._f("pdf_normal")(float(a), float(b), C.byref(e)); return v, e.value

    def log_odds_to_prob(self, l):
        return self._f("log_odds_to_prob")(float(l))

    def prob_to_log_odds(self, p):
        return self._f("prob_to_log_odds")(float(p))

    def _t(self, name, *arrs):
        out = np.empty(5)
        self._f(name)(*[_p(np.array(a, dtype=np.float64)) for a in arrs], _p(out))
        return out

    def make(self, pose): return self._t("transform_make", pose)
    def compose(self, a, b): return self._t("transform_compose", a, b)
    def inv(self, a): return self._t("transform_inv", a)
    def integrate_twist(self, a, tw): return self._t("transform_integrate_twist", a, tw)

    def apply(self, a, v):
        out = np.empty(2)
        self._f("transform_apply")(_p(np.array(a, dtype=np.float64)), _p(np.array(v, dtype=np.float64)), _p(out))
        return out

    # DiffDrive
    def dd_create(self, pose, wheel_base, wheel_radius):
        return C.c_void_p(self._f("dd_create")(_p(np.array(pose, dtype=np.float64)), float(wheel_base), float(wheel_radius)))

    def dd_destroy(self, d): self._f("dd_destroy")(d)

    def dd_twist_to_wheels(self, d, tw):
        out = np.empty(2); rc = self._f("dd_twist_to_wheels")(d, _p(np.array(tw, dtype=np.float64)), _p(out)); return out, rc

    def dd_wheels_to_twist(self, d, w):
        out = np.empty(3); self._f("dd_wheels_to_twist")(d, _p(np.array(w, dtype=np.float64)), _p(out)); return out

    def dd_update_odometry(self, d, left, right):
        out = np.empty(2); self._f("dd_update_odometry")(d, float(left), float(right), _p(out)); return out

    def dd_feedforward(self, d, tw): return self._f("dd_feedforward")(d, _p(np.array(tw, dtype=np.float64)))

    def dd_state(self, d):
        out = np.empty(7); self._f("dd_state")(d, _p(out)); return out

    def dd_arc_step(self, wheel_base, wheel_radius, dt, pose_xyt, wheels):
        """One plant step (wheelsToTwist * dt -> feedforward); returns the new (x, y, theta) and the error flag."""
        f = self._f("dd_arc_step")
        f.argtypes = [C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        pose = np.array(pose_xyt, dtype=np.float64)
        rc = f(float(wheel_base), float(wheel_radius), float(dt), _p(pose), _p(np.array(wheels, dtype=np.float64)))
        return pose, rc


# ---- particle filter (restatement only: the reference's particle_filter.cpp needs Eigen) ----------
class PfParams(C.Structure):
    _fields_ = [
        ("num_particles", C.c_int32), ("k", C.c_int32),
        ("srr", C.c_double), ("srt", C.c_double), ("str_", C.c_double), ("stt", C.c_double),
        ("motion_noise", C.c_double * 3), ("sample_range", C.c_double * 3),
        ("scan_min", C.c_double), ("scan_max", C.c_double), ("pose_min", C.c_double), ("pose_max", C.c_double),
        ("beam_min", C.c_float), ("beam_max", C.c_float), ("beam_delta", C.c_float), ("range_min", C.c_float),
        ("range_max", C.c_float), ("pad_", C.c_int32),
        ("z_hit", C.c_double), ("z_short", C.c_double), ("z_max", C.c_double), ("z_rand", C.c_double), ("sigma_hit", C.c_double),
        ("Trs", C.c_double * 3),
        ("resolution", C.c_double), ("xmin", C.c_double), ("xmax", C.c_double), ("ymin", C.c_double), ("ymax", C.c_double),
        ("pose0", C.c_double * 3),
    ]


class _Trace(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("sampled", "p_scan", "p_pose", "mu", "sigma", "eta", "new_pose", "weight_raw", "resample_idx")]


class PfStats(C.Structure):
    _fields_ = [("sum_w", C.c_double), ("sq_sum", C.c_double), ("neff", C.c_int32), ("resampled", C.c_int32),
                ("err", C.c_int32), ("normals_used", C.c_int32)]


# bmapping/launch/slam.launch:19-42
def pf_params(N=40, k=50, map_min=-2.0, map_max=2.0, beam_delta_deg=1.0, pose0=(0.0, 0.0, 0.0), **kw) -> PfParams:
    p = PfParams()
    p.num_particles, p.k = N, k
    p.srr, p.srt, p.str_, p.stt = 0.1, 0.2, 0.1, 0.2
    p.motion_noise[:] = [1e-10, 1e-10, 1e-10]
    p.sample_range[:] = [1e-10, 1e-8, 1e-8]
    p.scan_min, p.scan_max, p.pose_min, p.pose_max = 1.0, 20.0, 1.0, 10.0
    la = lds01_laser(beam_delta_deg)
    p.beam_min, p.beam_max, p.beam_delta, p.range_min, p.range_max = [float(v) for v in la]
    p.z_hit, p.z_short, p.z_max, p.z_rand, p.sigma_hit = [float(v) for v in MIX]
    p.Trs[:] = [0.0, 0.0, 0.0]
    p.resolution, p.xmin, p.xmax, p.ymin, p.ymax = 0.05, map_min, map_max, map_min, map_max
    p.pose0[:] = list(pose0)
    for key, v in kw.items():
        if isinstance(v, (list, tuple, np.ndarray)):
            getattr(p, key)[:] = list(v)
        else:
            setattr(p, key, v)
    return p


class PfAPI:
    def __init__(self, params: PfParams, exact_field=False, window=None):
        """exact_field: the likelihoods read the EXACT distance to the nearest occupied cell (what the device's default
        mode looks up) instead of the reference's brushfire — the checker of that mode, not the reference's behaviour.
        window = (i0, i1, j0, j1): only these rows / columns of every particle's map get storage (needs exact_field)."""
        self.L = lib()
        self.L.orc_pf_create.restype = C.c_void_p
        self.L.orc_pf_create_ex.restype = C.c_void_p
        self.L.orc_pf_grid.restype = C.c_void_p
        self.L.orc_pf_grid.argtypes = [C.c_void_p, C.c_int]
        self.p = params
        self.N, self.k = params.num_particles, params.k
        if exact_field or window is not None:
            win = None if window is None else np.array(window, dtype=np.int32)
            self.h = C.c_void_p(self.L.orc_pf_create_ex(C.byref(params), C.c_int(1 if exact_field else 0),
                                                        None if win is None else _p(win)))
            assert self.h, "a windowed oracle filter needs exact_field"
        else:
            self.h = C.c_void_p(self.L.orc_pf_create(C.byref(params)))

    def close(self):
        if self.h:
            self.L.orc_pf_destroy(self.h)
        self.h = None

    def normals_per_scan(self, icp_ok=True):
        return self.N * (3 * self.k + 3 if icp_ok else 3) + 1

    def slam(self, scan, u, cur_odom, prev_odom, icp_ok, T_icp, normals, trace=True):
        scan = np.ascontiguousarray(scan, dtype=np.float32)
        normals = np.ascontiguousarray(normals, dtype=np.float64)
        N, k = self.N, self.k
        tr = dict(sampled=np.zeros((N, k, 3)), p_scan=np.zeros((N, k)), p_pose=np.zeros((N, k)), mu=np.zeros((N, 3)),
                  sigma=np.zeros((N, 3, 3)), eta=np.zeros(N), new_pose=np.zeros((N, 3)), weight_raw=np.zeros(N),
                  resample_idx=np.full(N, -1, dtype=np.int32))
        t = _Trace(*[a.ctypes.data for a in tr.values()])
        st = PfStats()
        a = [np.array(v, dtype=np.float64) for v in (u, cur_odom, prev_odom, T_icp)]
        rc = self.L.orc_pf_slam(self.h, _p(scan), C.c_int(scan.size), _p(a[0]), _p(a[1]), _p(a[2]), C.c_int(1 if icp_ok else 0),
                                _p(a[3]), _p(normals), C.byref(t) if trace else None, C.byref(st))
        tr.update(rc=rc, sum_w=st.sum_w, sq_sum=st.sq_sum, neff=st.neff, resampled=st.resampled, normals_used=st.normals_used)
        return tr

    def particles(self):
        pose = np.empty((self.N, 3)); prev = np.empty((self.N, 3)); w = np.empty(self.N)
        self.L.orc_pf_get_particles(self.h, _p(pose), _p(prev), _p(w))
        return pose, prev, w

    def set_scan_matching(self, on, lstep=0.05, astep=0.05, iters=5):
        """N1 option (not the reference): refine T(pose)*T_icp per particle by hill climbing before sampling."""
        self.L.orc_pf_set_scan_matching(self.h, C.c_int(1 if on else 0), C.c_double(lstep), C.c_double(astep), C.c_int(iters))

    def scan_match_result(self):
        c = np.empty((self.N, 3)); sc = np.empty(self.N)
        self.L.orc_pf_get_scan_match(self.h, _p(c), _p(sc))
        return c, sc

    def set_particles(self, pose=None, prev=None, w=None):
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (pose, prev, w)]
        self.L.orc_pf_set_particles(self.h, *[None if a is None else _p(a) for a in arrs])

    def grid(self, p) -> GridAPI:
        return GridAPI("orc", _handle=self.L.orc_pf_grid(self.h, p), _borrowed=True)

    def best(self):
        return self.L.orc_pf_best(self.h)


def exact_edt_codes(occ: np.ndarray, radius: int, prev: np.ndarray) -> np.ndarray:
    xs, ys = occ.shape
    occ8 = np.ascontiguousarray(occ, dtype=np.uint8); pv = np.ascontiguousarray(prev, dtype=np.uint16)
    out = np.empty((xs, ys), dtype=np.uint16)
    lib().orc_exact_edt_codes(C.c_int(xs), C.c_int(ys), _p(occ8), C.c_int(radius), _p(pv), _p(out))
    return out


# ---- synthetic world: axis-aligned room, SURVEY.md 8-d ---------------------------------------------
def room_scan(pose, n_beams=360, beam_delta_deg=1.0, walls=(-3.0, 3.0, -2.5, 2.5), noise_sigma=0.01, rng=None,
              range_max=3.5):
    """Ranges (float32) a lidar at pose=(theta,x,y) sees in a rectangular room; ranges >= range_max are
    kept (the scanner gates them out itself)."""
    th, x, y = pose
    xmin, xmax, ymin, ymax = walls
    ang = th + np.deg2rad(beam_delta_deg) * np.arange(n_beams)
    c, s = np.cos(ang), np.sin(ang)
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = np.where(c > 0, (xmax - x) / c, np.where(c < 0, (xmin - x) / c, np.inf))
        ty = np.where(s > 0, (ymax - y) / s, np.where(s < 0, (ymin - y) / s, np.inf))
    r = np.minimum(tx, ty)
    if rng is not None and noise_sigma > 0:
        r = r + rng.normal(0.0, noise_sigma, size=r.shape)
    return r.astype(np.float32)
