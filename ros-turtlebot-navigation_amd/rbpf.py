"""Python mirror of bmapping::ParticleFilter over the C-ABI (tests / smoke / bench plumbing).

Same method names and argument meaning as the reference's class surface
(bmapping/include/bmapping/particle_filter.hpp:88-146): SLAM(scan, u, cur_odom, prev_odom),
getRobotState(), newMap().  ICP runs on the host before SLAM exactly where particle_filter.cpp:153
calls it, so its result (icp_ok, T_icp) is passed in, as are the standard-normal draws.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


def default_params(N=40, k=50, map_min=-2.0, map_max=2.0, beam_delta_deg=1.0, pose0=(0.0, 0.0, 0.0), device=-1,
                   **kw) -> "capi.RbpfParams":
    """The shipped configuration: bmapping/launch/slam.launch:19-42, config/LDS_01_lidar.yaml
    (degrees converted like turtle_mapping_node.cpp:300-302, then narrowed to float like
    LaserProperties, sensor_model.hpp:23-27)."""
    p = capi.RbpfParams()
    p.num_particles, p.num_samples_mode = N, k
    p.srr, p.srt, p.str_, p.stt = 0.1, 0.2, 0.1, 0.2
    p.motion_noise[:] = [1e-10, 1e-10, 1e-10]
    p.sample_range[:] = [1e-10, 1e-8, 1e-8]
    p.scan_likelihood_min, p.scan_likelihood_max = 1.0, 20.0
    p.pose_likelihood_min, p.pose_likelihood_max = 1.0, 10.0
    d2r = np.pi / 180.0
    p.beam_min, p.beam_max, p.beam_delta = 0.0, float(np.float32(360.0 * d2r)), float(np.float32(beam_delta_deg * d2r))
    p.range_min, p.range_max = 0.12, 3.5
    p.z_hit, p.z_short, p.z_max, p.z_rand, p.sigma_hit = 0.95, 0.0, 0.04, 0.01, 0.5
    p.Trs[:] = [0.0, 0.0, 0.0]
    p.resolution, p.xmin, p.xmax, p.ymin, p.ymax = 0.05, map_min, map_max, map_min, map_max
    p.pose0[:] = list(pose0)
    p.device = device
    for key, v in kw.items():
        if isinstance(v, (list, tuple, np.ndarray)):
            getattr(p, key)[:] = list(v)
        else:
            setattr(p, key, v)
    return p


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


class ParticleFilter:
    """bmapping::ParticleFilter on one MI355X."""

    def __init__(self, params: "capi.RbpfParams", pool_bytes: int = 0, df_mode: str | None = None):
        """pool_bytes: budget of the log-odds tile pool (0 = default policy, tbnav_rbpf.h); df_mode: one of
        "query" (default) / "window" / "full" / "reference" (TBNAV_RBPF_OPT_DF_MODE)."""
        self._L = capi.lib()
        self.params = params
        self._h = C.c_void_p()
        capi.check(self._L.tbnav_rbpf_create_pool(C.byref(params), int(pool_bytes), C.byref(self._h)), "tbnav_rbpf_create")
        if df_mode is not None:
            capi.check(self._L.tbnav_rbpf_set_option(self._h, capi.RBPF_OPT_DF_MODE, capi.RBPF_DF[df_mode]), "set_option(df_mode)")
        xs, ys = C.c_int32(), C.c_int32()
        capi.check(self._L.tbnav_rbpf_grid_size(self._h, C.byref(xs), C.byref(ys)), "grid_size")
        self.xsize, self.ysize = xs.value, ys.value
        self.G = self.xsize * self.ysize
        self.N, self.k = params.num_particles, params.num_samples_mode
        self.last_stats = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value and getattr(self, "_owned", True):
            self._L.tbnav_rbpf_destroy(self._h)
        self._h = C.c_void_p()

    __del__ = close

    def attachComm(self, comm):
        """SLAM() of this handle becomes this rank's part of the sharded scan, exchanged inside the library (rtn_amd.comm.Comm;
        None detaches).  The caller sets the weights to 1 / N_global first (setParticles)."""
        capi.check(self._L.tbnav_rbpf_attach_comm(self._h, comm._h if comm is not None else None), "tbnav_rbpf_attach_comm")
        self._comm = comm

    def numNormals(self, icp_ok=True) -> int:
        return int(self._L.tbnav_rbpf_num_normals(self._h, 1 if icp_ok else 0))

    # ---- reference surface ----
    def SLAM(self, scan, u, cur_odom, prev_odom, icp_ok, T_icp, normals, local_only=False, check=True):
        """u = (w, vx, vy); odometry and T_icp as (theta, x, y).  Returns the stats struct; raises
        TbnavError with the reference's exception text when the reference would have thrown."""
        scan = np.ascontiguousarray(scan, dtype=np.float32)
        if normals is not None:
            normals = np.ascontiguousarray(normals, dtype=np.float64)
            assert normals.size >= self.numNormals(icp_ok)
        st = capi.RbpfStats()
        fn = self._L.tbnav_rbpf_slam_local if local_only else self._L.tbnav_rbpf_slam
        rc = fn(self._h, scan.ctypes.data, scan.size, _d3(u), _d3(cur_odom), _d3(prev_odom), 1 if icp_ok else 0,
                _d3(T_icp), normals.ctypes.data if normals is not None else None, C.byref(st))
        self.last_stats = st
        if check:
            capi.check(rc, "tbnav_rbpf_slam")
        return st

    def SLAMBatch(self, scans, u, odom, T_icp, icp_ok=None, check=True):
        """Replay of a logged run (tbnav_rbpf_slam_batch): scans [n][n_beams], u [n][3], odom [n + 1][3] (odom[s] = prev,
        odom[s + 1] = cur), T_icp [n][3]; device noise.  Returns the list of per-scan stats (check=False: also when a
        scan's status stopped the replay — the stats of the scans after it are zero)."""
        scans = np.ascontiguousarray(scans, dtype=np.float32)
        n, nb = scans.shape
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(n, 3)
        odom = np.ascontiguousarray(odom, dtype=np.float64).reshape(n + 1, 3)
        T_icp = np.ascontiguousarray(T_icp, dtype=np.float64).reshape(n, 3)
        ok = None if icp_ok is None else np.ascontiguousarray(icp_ok, dtype=np.int32)
        out = (capi.RbpfStats * n)()
        rc = self._L.tbnav_rbpf_slam_batch(self._h, scans.ctypes.data, nb, n, u.ctypes.data, odom.ctypes.data,
                                           None if ok is None else ok.ctypes.data, T_icp.ctypes.data, C.cast(out, C.c_void_p))
        if check:
            capi.check(rc, "tbnav_rbpf_slam_batch")
        self.last_stats = out[n - 1]
        return list(out)

    def setSeed(self, seed: int):
        capi.check(self._L.tbnav_rbpf_set_seed(self._h, seed), "set_seed")

    def setRngShard(self, first_particle: int, particles_global: int):
        """This handle's particles are [first_particle, first_particle + N) of an ensemble of particles_global: the device noise
        source draws the ensemble's normals for them (ranks sharing a seed draw disjoint normals, the unsharded filter's)."""
        capi.check(self._L.tbnav_rbpf_set_rng_shard(self._h, first_particle, particles_global), "set_rng_shard")

    def lastNormals(self, n: int) -> np.ndarray:
        out = np.empty(n)
        capi.check(self._L.tbnav_rbpf_get_normals(self._h, out.ctypes.data, n), "get_normals")
        return out

    def getRobotState(self):
        pose = (C.c_double * 3)(); idx = C.c_int32()
        capi.check(self._L.tbnav_rbpf_best_state(self._h, pose, C.byref(idx)), "best_state")
        return (pose[0], pose[1], pose[2]), idx.value

    def newMap(self) -> np.ndarray:
        m = np.empty(self.G, dtype=np.int8)
        capi.check(self._L.tbnav_rbpf_best_map(self._h, m.ctypes.data), "best_map")
        return m

    # ---- state access ----
    def particles(self):
        pose = np.empty((self.N, 3)); prev = np.empty((self.N, 3)); w = np.empty(self.N)
        capi.check(self._L.tbnav_rbpf_get_particles(self._h, pose.ctypes.data, prev.ctypes.data, w.ctypes.data), "get_particles")
        return pose, prev, w

    def setParticles(self, pose=None, prev=None, w=None):
        a = [None if v is None else np.ascontiguousarray(v, dtype=np.float64) for v in (pose, prev, w)]
        capi.check(self._L.tbnav_rbpf_set_particles(self._h, *[None if v is None else v.ctypes.data for v in a]), "set_particles")

    def logOdds(self, p) -> np.ndarray:
        out = np.empty(self.G)
        capi.check(self._L.tbnav_rbpf_get_log_odds(self._h, p, out.ctypes.data), "get_log_odds")
        return out

    def setLogOdds(self, p, lo):
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        capi.check(self._L.tbnav_rbpf_set_log_odds(self._h, p, lo.ctypes.data), "set_log_odds")

    def occDist(self, p) -> np.ndarray:
        out = np.empty(self.G)
        capi.check(self._L.tbnav_rbpf_get_occ_dist(self._h, p, out.ctypes.data), "get_occ_dist")
        return out

    def setOccDist(self, p, od):
        od = np.ascontiguousarray(od, dtype=np.float64)
        capi.check(self._L.tbnav_rbpf_set_occ_dist(self._h, p, od.ctypes.data), "set_occ_dist")

    def distCode(self, p) -> np.ndarray:
        out = np.empty(self.G, dtype=np.uint16)
        capi.check(self._L.tbnav_rbpf_get_dist_code(self._h, p, out.ctypes.data), "get_dist_code")
        return out

    def occupiedCount(self) -> np.ndarray:
        out = np.empty(self.N, dtype=np.int32)
        capi.check(self._L.tbnav_rbpf_get_occupied_count(self._h, out.ctypes.data), "get_occupied_count")
        return out

    def trace(self) -> dict:
        N, k = self.N, self.k
        t = dict(sampled=np.empty((N, k, 3)), p_scan=np.empty((N, k)), p_pose=np.empty((N, k)), mu=np.empty((N, 3)),
                 sigma=np.empty((N, 3, 3)), eta=np.empty(N), new_pose=np.empty((N, 3)), weight_raw=np.empty(N),
                 resample_idx=np.empty(N, dtype=np.int32))
        capi.check(self._L.tbnav_rbpf_get_trace(self._h, *[a.ctypes.data for a in t.values()]), "get_trace")
        return t

    def setScanMatching(self, on: bool = True, lstep: float = 0.05, astep: float = 0.05, iterations: int = 5):
        """Option, not the reference: every particle refines T(pose) * T_icp against its own map (hill climbing on
        the likelihood field) before the samples are drawn round it."""
        capi.check(self._L.tbnav_rbpf_set_scan_matching(self._h, 1 if on else 0, lstep, astep, iterations), "set_scan_matching")

    def scanMatch(self):
        """Matched poses [N][3] (theta, x, y) and scores [N] of the last SLAM call with scan matching on."""
        c = np.empty((self.N, 3)); sc = np.empty(self.N)
        capi.check(self._L.tbnav_rbpf_get_scan_match(self._h, c.ctypes.data, sc.ctypes.data), "get_scan_match")
        return c, sc

    def setOption(self, option: int, value: int):
        capi.check(self._L.tbnav_rbpf_set_option(self._h, option, value), "set_option")

    def poolStats(self):
        """(tiles the pool holds, tiles free, bytes per tile)."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        capi.check(self._L.tbnav_rbpf_pool_stats(self._h, C.byref(a), C.byref(b), C.byref(c)), "pool_stats")
        return a.value, b.value, c.value

    def scanCounts(self, reset: bool = True):
        """(cell updates, distinct cells written) summed since the last reset; needs setOption(RBPF_OPT_COUNT_CELLS, 1)."""
        a, b = C.c_uint64(), C.c_uint64()
        capi.check(self._L.tbnav_rbpf_scan_counts(self._h, C.byref(a), C.byref(b), 1 if reset else 0), "scan_counts")
        return a.value, b.value

    def referenceFieldCounts(self):
        """Reference-field mode: (distinct states the particles hold, brushfires run by the last scan, by all scans) —
        tbnav_rbpf_reference_field_counts."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int64()
        capi.check(self._L.tbnav_rbpf_reference_field_counts(self._h, C.byref(a), C.byref(b), C.byref(c)), "reference_field_counts")
        return a.value, b.value, c.value

    def referenceFieldStats(self):
        """Reference-field mode: the lazy brushfire's counters (tbnav_rbpf_reference_field_stats) as a dict."""
        out = (C.c_int64 * 18)()
        capi.check(self._L.tbnav_rbpf_reference_field_stats(self._h, out), "reference_field_stats")
        keys = ("passes", "iterations", "states_resumed", "passes_completed", "lineages_replayed", "generations_replayed", "history_bytes", "proposals_rerun",
                "us_logs", "us_step", "us_resample", "us_flush", "us_before_propose", "us_settle_look", "us_step_grouping", "us_step_release",
                "us_threads_busy", "host_threads_last_scan")
        return dict(zip(keys, (int(v) for v in out)))

    def setTiming(self, on: bool = True):
        """Record HIP events round the kernels of the following SLAM calls (they cost device time: off by default)."""
        capi.check(self._L.tbnav_rbpf_set_timing(self._h, 1 if on else 0), "set_timing")

    def kernelMs(self):
        """Per-kernel durations of the last SLAM call; zeros unless setTiming(True) was called before it."""
        ms = (C.c_float * 6)()
        capi.check(self._L.tbnav_rbpf_last_kernel_ms(self._h, ms), "last_kernel_ms")
        return dict(zip(("propose", "raycast", "occupancy", "edt", "normalize", "gather"), [float(x) for x in ms]))


    def lastKernelNames(self):
        """(propose, raycast, raycast workgroups): the instantiations the last launches were, as rocprofv3 prints them."""
        a, b, n = C.create_string_buffer(64), C.create_string_buffer(64), C.c_int32()
        capi.check(self._L.tbnav_rbpf_last_kernel_names(self._h, a, 64, b, 64, C.byref(n)), "last_kernel_names")
        return a.value.decode(), b.value.decode(), int(n.value)


    def raycastBoxCells(self):
        """(cells the particles' boxes needed lately, cells of the last launch's LDS array) of rbpf_raycast_box."""
        a, b = C.c_int32(), C.c_int32()
        capi.check(self._L.tbnav_rbpf_raycast_box_cells(self._h, C.byref(a), C.byref(b)), "raycast_box_cells")
        return int(a.value), int(b.value)


class ParticleFilterGroup:
    """tbnav_rbpf_group: ONE process driving the filter over several devices (what bmapping::ParticleFilter(..., n_gpus) holds).
    params.num_particles is the ensemble's N; devices may repeat (members sharing a device exchange by copies, not RCCL)."""

    def __init__(self, params: "capi.RbpfParams", devices, pool_bytes_per_member: int = 0):
        self._L = capi.lib()
        self.params = params
        self._h = C.c_void_p()
        d = np.ascontiguousarray(devices, dtype=np.int32)
        capi.check(self._L.tbnav_rbpf_group_create(C.byref(params), len(d), d.ctypes.data, int(pool_bytes_per_member), C.byref(self._h)),
                   "tbnav_rbpf_group_create")
        self.n = len(d)
        self.N, self.k = params.num_particles, params.num_samples_mode
        self.n_local = self.N // self.n
        m0 = self.member(0)
        self.xsize, self.ysize, self.G = m0.xsize, m0.ysize, m0.G

    def member(self, r: int) -> ParticleFilter:
        """A borrowed view of shard r (parity hooks: particles(), logOdds(), trace() ...)."""
        h = C.c_void_p()
        capi.check(self._L.tbnav_rbpf_group_member(self._h, r, C.byref(h)), "tbnav_rbpf_group_member")
        m = ParticleFilter.__new__(ParticleFilter)
        m._L, m._h, m._owned, m.params = self._L, h, False, self.params
        xs, ys = C.c_int32(), C.c_int32()
        capi.check(self._L.tbnav_rbpf_grid_size(h, C.byref(xs), C.byref(ys)), "grid_size")
        m.xsize, m.ysize = xs.value, ys.value
        m.G = m.xsize * m.ysize
        m.N, m.k, m.last_stats = self.N // self.n, self.k, None
        return m

    def numNormals(self, icp_ok=True) -> int:
        return int(self._L.tbnav_rbpf_group_num_normals(self._h, 1 if icp_ok else 0))

    def setSeed(self, seed: int):
        capi.check(self._L.tbnav_rbpf_group_set_seed(self._h, seed), "group_set_seed")

    def setOption(self, option: int, value: int):
        capi.check(self._L.tbnav_rbpf_group_set_option(self._h, option, value), "group_set_option")

    def SLAM(self, scan, u, cur_odom, prev_odom, icp_ok, T_icp, normals, check=True):
        """normals: the ENSEMBLE's draw stream (numNormals values, the reference's order) or None (device noise)."""
        scan = np.ascontiguousarray(scan, dtype=np.float32)
        if normals is not None:
            normals = np.ascontiguousarray(normals, dtype=np.float64)
            assert normals.size >= self.numNormals(icp_ok)
        st = capi.RbpfStats()
        rc = self._L.tbnav_rbpf_group_slam(self._h, scan.ctypes.data, scan.size, _d3(u), _d3(cur_odom), _d3(prev_odom), 1 if icp_ok else 0,
                                           _d3(T_icp), normals.ctypes.data if normals is not None else None, C.byref(st))
        if check:
            capi.check(rc, "tbnav_rbpf_group_slam")
        return st

    def particles(self):
        parts = [self.member(r).particles() for r in range(self.n)]
        return tuple(np.concatenate([p[i] for p in parts]) for i in range(3))

    def setParticles(self, pose=None, prev=None, w=None):
        nl = self.n_local
        for r in range(self.n):
            sl = slice(r * nl, (r + 1) * nl)
            self.member(r).setParticles(None if pose is None else np.asarray(pose)[sl], None if prev is None else np.asarray(prev)[sl],
                                        None if w is None else np.asarray(w)[sl])

    def logOdds(self, p: int) -> np.ndarray:
        return self.member(p // self.n_local).logOdds(p % self.n_local)

    def getRobotState(self):
        pose = (C.c_double * 3)(); idx = C.c_int32()
        capi.check(self._L.tbnav_rbpf_group_best_state(self._h, pose, C.byref(idx)), "group_best_state")
        return (pose[0], pose[1], pose[2]), idx.value

    def newMap(self) -> np.ndarray:
        m = np.empty(self.G, dtype=np.int8)
        capi.check(self._L.tbnav_rbpf_group_best_map(self._h, m.ctypes.data), "group_best_map")
        return m

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.tbnav_rbpf_group_destroy(self._h)
        self._h = C.c_void_p()

    __del__ = close


def resample_global(weights_all: np.ndarray, z: float):
    """normalizeWeights + effectiveParticles + lowVarianceResampling on the global weight vector
    (host, sequential order).  Returns (parents, normalised weights, stats)."""
    w = np.ascontiguousarray(weights_all, dtype=np.float64)
    parents = np.empty(w.size, dtype=np.int32); wn = np.empty(w.size)
    st = capi.RbpfStats()
    capi.check(capi.lib().tbnav_rbpf_resample_global(w.ctypes.data, w.size, float(z), parents.ctypes.data,
                                                     wn.ctypes.data, C.byref(st)), "resample_global")
    return parents, wn, st
