"""ctypes binding of the C-ABI declared in include/tbnav_*.h (libtbnav_hip.so).

This is plumbing for tests, smoke() and bench.py: it loads the in-tree shared library and exposes
the entry points with typed signatures.  There is NO fallback: if the library is missing or a call
fails, it raises.  Nothing here imports or calls oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtbnav_hip.so")
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

TBNAV_MPPI_REC = 8

# tbnav_status.h
OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_OUT_OF_WORLD, ERR_ETA_ZERO, ERR_PDF_VARIANCE, \
    ERR_BRESENHAM, ERR_UNSUPPORTED, ERR_POOL_EXHAUSTED = range(10)

# tbnav_mppi.h options
MPPI_OPT_KERNEL, MPPI_OPT_TRIG, MPPI_OPT_NO_LDS_STAGING, MPPI_OPT_KEEP_J, MPPI_OPT_REG_TAIL, MPPI_OPT_BATCH_GRAPH, MPPI_OPT_PREFIX_FORM = 1, 2, 3, 4, 5, 6, 7
MPPI_OPT_DIRECT_EXCHANGE, MPPI_OPT_SAMPLER, MPPI_OPT_FAULT_INJECT, MPPI_OPT_WIDE_COMBINE = 8, 9, 10, 11

# tbnav_rbpf.h options
RBPF_OPT_DF_MODE, RBPF_OPT_RAYCAST_ORDERED, RBPF_OPT_RAYCAST_THREADS, RBPF_OPT_COUNT_CELLS, RBPF_OPT_RAYCAST_FORM = 1, 2, 3, 4, 5
RBPF_OPT_RAYCAST_BAND_ROWS = 6
RBPF_OPT_RAYCAST_ADAPT = 9
RBPF_OPT_RAYCAST_CELL16 = 10
RBPF_OPT_NOISE_IN_KERNEL = 11
RBPF_OPT_BATCH_PIPELINE = 7
RBPF_OPT_HOST_THREADS = 8
RBPF_OPT_REF_REACH = 12
RBPF_DF = {"full": 0, "window": 1, "query": 2, "reference": 3}


class TbnavError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: status {status}" + (f" ({detail})" if detail else ""))


class MppiParams(C.Structure):
    """tbnav_mppi_params (include/tbnav_mppi.h)."""
    _fields_ = [
        ("wheel_radius", C.c_double), ("wheel_base", C.c_double), ("lam", C.c_double),
        ("max_wheel_vel", C.c_double), ("ul_var", C.c_double), ("ur_var", C.c_double),
        ("horizon", C.c_double), ("dt", C.c_double),
        ("Q", C.c_double * 3), ("R", C.c_double * 2), ("P1", C.c_double * 3),
        ("rollouts", C.c_int32), ("device", C.c_int32),
    ]


class RbpfParams(C.Structure):
    """tbnav_rbpf_params (include/tbnav_rbpf.h)."""
    _fields_ = [
        ("num_particles", C.c_int32), ("num_samples_mode", C.c_int32),
        ("srr", C.c_double), ("srt", C.c_double), ("str_", C.c_double), ("stt", C.c_double),
        ("motion_noise", C.c_double * 3), ("sample_range", C.c_double * 3),
        ("scan_likelihood_min", C.c_double), ("scan_likelihood_max", C.c_double),
        ("pose_likelihood_min", C.c_double), ("pose_likelihood_max", C.c_double),
        ("beam_min", C.c_float), ("beam_max", C.c_float), ("beam_delta", C.c_float),
        ("range_min", C.c_float), ("range_max", C.c_float), ("device", C.c_int32),
        ("z_hit", C.c_double), ("z_short", C.c_double), ("z_max", C.c_double), ("z_rand", C.c_double),
        ("sigma_hit", C.c_double),
        ("Trs", C.c_double * 3),
        ("resolution", C.c_double), ("xmin", C.c_double), ("xmax", C.c_double), ("ymin", C.c_double),
        ("ymax", C.c_double),
        ("pose0", C.c_double * 3),
    ]


class RbpfStats(C.Structure):
    """tbnav_rbpf_stats."""
    _fields_ = [("sum_w", C.c_double), ("sq_sum", C.c_double), ("neff", C.c_int32), ("resampled", C.c_int32),
                ("status", C.c_int32), ("n_valid_beams", C.c_int32)]


_lib = None


def declared_symbols() -> list[str]:
    """Every function name declared in include/tbnav_*.h (used by the symbol-export test)."""
    names: list[str] = []
    for fn in sorted(os.listdir(INCLUDE_DIR)):
        if not (fn.startswith("tbnav_") and fn.endswith(".h")):
            continue
        text = open(os.path.join(INCLUDE_DIR, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(tbnav_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def lib() -> C.CDLL:
    """Load libtbnav_hip.so (once).  Raises if it has not been built — never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch bundles its own HIP runtime (torch/lib/libamdhip64.so) and must be the first to load
    # one: if /opt/rocm's copy (our DT_NEEDED) is mapped first, torch later maps a second runtime
    # and sees "No HIP GPUs".  With torch first, our NEEDED resolves to the already-loaded SONAME
    # and the process has ONE runtime, so torch streams/pointers are valid in our kernels.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the HIP path has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    dp, vp, i32, u64, dbl = C.POINTER(C.c_double), C.c_void_p, C.c_int32, C.c_uint64, C.c_double
    sig = {
        "tbnav_status_string": (C.c_char_p, [C.c_int]),
        "tbnav_last_hip_error": (C.c_char_p, []),
        "tbnav_device_count": (C.c_int, []),
        # MPPI
        "tbnav_mppi_create": (C.c_int, [C.POINTER(MppiParams), C.POINTER(vp)]),
        "tbnav_mppi_destroy": (None, [vp]),
        "tbnav_mppi_steps": (C.c_int, [vp]),
        "tbnav_mppi_rollouts": (C.c_int, [vp]),
        "tbnav_mppi_rollout_variant": (C.c_int, [vp]),
        "tbnav_mppi_streaming_form": (C.c_int, [vp]),
        "tbnav_mppi_graph_replayed_ticks": (C.c_int64, [vp]),
        "tbnav_mppi_set_dynamics": (C.c_int, [vp, i32]),
        "tbnav_mppi_set_option": (C.c_int, [vp, i32, i32]),
        "tbnav_mppi_set_rng_shard": (C.c_int, [vp, u64, u64]),
        "tbnav_mppi_records_per_step": (C.c_int, [vp]),
        "tbnav_mppi_set_initial_controls": (C.c_int, [vp, dbl, dbl]),
        "tbnav_mppi_set_waypoint": (C.c_int, [vp, dbl, dbl, dbl]),
        "tbnav_mppi_get_controls": (C.c_int, [vp, vp]),
        "tbnav_mppi_set_controls": (C.c_int, [vp, vp]),
        "tbnav_mppi_new_controls": (C.c_int, [vp, dp, vp, dp]),
        "tbnav_mppi_new_controls_dev": (C.c_int, [vp, dp, vp, vp, vp, dp]),
        "tbnav_mppi_enqueue_dev": (C.c_int, [vp, dp, vp, vp, vp]),
        "tbnav_mppi_last_controls": (C.c_int, [vp, vp, dp]),
        "tbnav_mppi_sample_noise": (C.c_int, [vp, u64, u64, vp]),
        "tbnav_mppi_enqueue_rng": (C.c_int, [vp, dp, u64, u64, vp]),
        "tbnav_mppi_enqueue_rng_batch": (C.c_int, [vp, dp, i32, u64, u64, i32, vp]),
        "tbnav_mppi_new_controls_rng": (C.c_int, [vp, dp, u64, u64, vp, dp]),
        "tbnav_mppi_get_noise": (C.c_int, [vp, vp, vp]),
        "tbnav_mppi_shard_partials": (C.c_int, [vp, dp, vp, vp, vp, vp]),
        "tbnav_mppi_shard_combine": (C.c_int, [vp, vp, i32, vp]),
        "tbnav_mppi_shard_partials_rng": (C.c_int, [vp, dp, u64, u64, vp, vp]),
        "tbnav_mppi_get_cost_to_go": (C.c_int, [vp, vp]),
        "tbnav_mppi_debug_sincos": (C.c_int, [vp, i32, vp, vp]),
        "tbnav_mppi_debug_div_lambda": (C.c_int, [vp, i32, dbl, vp, C.POINTER(C.c_int32)]),
        "tbnav_mppi_profile_tick": (C.c_int, [vp, dp, vp, vp, vp, C.POINTER(C.c_float)]),
        "tbnav_mppi_profile_kernels": (C.c_int, [vp, dp, vp, vp, vp, i32, C.POINTER(C.c_float)]),
        "tbnav_mppi_profile_kernels_rng": (C.c_int, [vp, dp, C.c_uint64, C.c_uint64, vp, i32, C.POINTER(C.c_float)]),
        "tbnav_mppi_last_kernel_names": (C.c_int, [vp, C.c_char_p, i32, C.c_char_p, i32]),
        # communicators (include/tbnav_comm.h) and the sharded MPPI tick
        "tbnav_comm_unique_id": (C.c_int, [vp]),
        "tbnav_comm_unique_id_ipc": (C.c_int, [vp]),
        "tbnav_comm_create": (C.c_int, [vp, i32, i32, i32, C.POINTER(vp)]),
        "tbnav_comm_create_local": (C.c_int, [i32, vp, vp]),
        "tbnav_comm_destroy": (None, [vp]),
        "tbnav_comm_selftest": (C.c_int, [vp, u64]),
        "tbnav_comm_rank": (C.c_int, [vp]),
        "tbnav_comm_size": (C.c_int, [vp]),
        "tbnav_comm_device": (C.c_int, [vp]),
        "tbnav_comm_uses_rccl": (C.c_int, [vp]),
        "tbnav_mppi_attach_comm": (C.c_int, [vp, vp]),
        "tbnav_mppi_exchange_kind": (C.c_int, [vp]),
        "tbnav_mppi_exchange_probe": (C.c_int, [vp, i32, vp, C.POINTER(C.c_double)]),
        "tbnav_mppi_group_create": (C.c_int, [C.POINTER(MppiParams), i32, vp, C.POINTER(vp)]),
        "tbnav_mppi_group_destroy": (None, [vp]),
        "tbnav_mppi_group_size": (C.c_int, [vp]),
        "tbnav_mppi_group_member": (C.c_int, [vp, i32, C.POINTER(vp)]),
        "tbnav_mppi_group_set_waypoint": (C.c_int, [vp, dbl, dbl, dbl]),
        "tbnav_mppi_group_set_initial_controls": (C.c_int, [vp, dbl, dbl]),
        "tbnav_mppi_group_set_controls": (C.c_int, [vp, vp]),
        "tbnav_mppi_group_get_controls": (C.c_int, [vp, vp]),
        "tbnav_mppi_group_set_dynamics": (C.c_int, [vp, i32]),
        "tbnav_mppi_group_set_option": (C.c_int, [vp, i32, i32]),
        "tbnav_mppi_group_new_controls": (C.c_int, [vp, dp, vp, dp]),
        "tbnav_mppi_group_new_controls_rng": (C.c_int, [vp, dp, u64, u64, dp]),
        "tbnav_mppi_group_enqueue_rng": (C.c_int, [vp, dp, u64, u64]),
        "tbnav_mppi_group_enqueue_rng_batch": (C.c_int, [vp, vp, i32, u64, u64, i32]),
        "tbnav_mppi_group_last_controls": (C.c_int, [vp, dp]),
        "tbnav_mppi_group_synchronize": (C.c_int, [vp]),
        # RBPF
        "tbnav_rbpf_attach_comm": (C.c_int, [vp, vp]),
        "tbnav_rbpf_group_create": (C.c_int, [C.POINTER(RbpfParams), i32, vp, u64, C.POINTER(vp)]),
        "tbnav_rbpf_group_destroy": (None, [vp]),
        "tbnav_rbpf_group_size": (C.c_int, [vp]),
        "tbnav_rbpf_group_member": (C.c_int, [vp, i32, C.POINTER(vp)]),
        "tbnav_rbpf_group_set_seed": (C.c_int, [vp, u64]),
        "tbnav_rbpf_group_set_option": (C.c_int, [vp, i32, i32]),
        "tbnav_rbpf_group_num_normals": (C.c_int64, [vp, i32]),
        "tbnav_rbpf_group_slam": (C.c_int, [vp, vp, i32, dp, dp, dp, i32, dp, vp, C.POINTER(RbpfStats)]),
        "tbnav_rbpf_group_best_state": (C.c_int, [vp, dp, C.POINTER(i32)]),
        "tbnav_rbpf_group_best_map": (C.c_int, [vp, vp]),
        "tbnav_rbpf_create": (C.c_int, [C.POINTER(RbpfParams), C.POINTER(vp)]),
        "tbnav_rbpf_create_pool": (C.c_int, [C.POINTER(RbpfParams), u64, C.POINTER(vp)]),
        "tbnav_rbpf_pool_stats": (C.c_int, [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
        "tbnav_rbpf_set_option": (C.c_int, [vp, i32, i32]),
        "tbnav_rbpf_integrate_scan": (C.c_int, [vp, i32, vp, i32, dp]),
        "tbnav_rbpf_likelihood": (C.c_int, [vp, i32, vp, i32, dp, dp]),
        "tbnav_rbpf_particle_map": (C.c_int, [vp, i32, vp]),
        "tbnav_rbpf_scan_counts": (C.c_int, [vp, C.POINTER(u64), C.POINTER(u64), i32]),
        "tbnav_rbpf_reference_field_counts": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_int64)]),
        "tbnav_rbpf_reference_field_stats": (C.c_int, [vp, C.POINTER(C.c_int64)]),
        "tbnav_rbpf_destroy": (None, [vp]),
        "tbnav_rbpf_grid_size": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32)]),
        "tbnav_rbpf_num_normals": (C.c_int64, [vp, i32]),
        "tbnav_rbpf_set_seed": (C.c_int, [vp, u64]),
        "tbnav_rbpf_set_rng_shard": (C.c_int, [vp, u64, u64]),
        "tbnav_rbpf_get_normals": (C.c_int, [vp, vp, C.c_int64]),
        "tbnav_rbpf_slam": (C.c_int, [vp, vp, i32, dp, dp, dp, i32, dp, vp, C.POINTER(RbpfStats)]),
        "tbnav_rbpf_slam_local": (C.c_int, [vp, vp, i32, dp, dp, dp, i32, dp, vp, C.POINTER(RbpfStats)]),
        "tbnav_rbpf_slam_batch": (C.c_int, [vp, vp, i32, i32, vp, vp, vp, vp, vp]),
        "tbnav_rbpf_resample_global": (C.c_int, [vp, C.c_int64, dbl, vp, vp, C.POINTER(RbpfStats)]),
        "tbnav_rbpf_add_repeated": (C.c_int, [vp, vp, vp, vp, C.c_int64]),
        "tbnav_rbpf_pool_selftest": (C.c_int, [C.c_uint32, i32, i32, i32, i32, vp, vp, vp]),
        "tbnav_rbpf_gather_local": (C.c_int, [vp, vp]),
        "tbnav_rbpf_copy_weights_dev": (C.c_int, [vp, vp]),
        "tbnav_rbpf_resample_global_dev": (C.c_int, [vp, vp, C.c_int64, C.c_int64, dbl, vp, C.POINTER(RbpfStats)]),
        "tbnav_rbpf_set_weights_from_global_dev": (C.c_int, [vp, vp]),
        "tbnav_rbpf_export_size": (C.c_int, [vp, i32, C.POINTER(u64)]),
        "tbnav_rbpf_export_particle_dev": (C.c_int, [vp, i32, vp, u64, C.POINTER(u64)]),
        "tbnav_rbpf_import_particle_dev": (C.c_int, [vp, i32, vp, u64]),
        "tbnav_rbpf_export_batch_sizes": (C.c_int, [vp, i32, vp, vp]),
        "tbnav_rbpf_export_batch_dev": (C.c_int, [vp, i32, vp, vp, u64, vp]),
        "tbnav_rbpf_import_batch_dev": (C.c_int, [vp, i32, vp, vp, u64, vp]),
        "tbnav_rbpf_copy_particle": (C.c_int, [vp, i32, vp, i32]),
        "tbnav_rbpf_best_state": (C.c_int, [vp, dp, C.POINTER(i32)]),
        "tbnav_rbpf_best_map": (C.c_int, [vp, vp]),
        "tbnav_rbpf_get_particles": (C.c_int, [vp, vp, vp, vp]),
        "tbnav_rbpf_set_particles": (C.c_int, [vp, vp, vp, vp]),
        "tbnav_rbpf_get_log_odds": (C.c_int, [vp, i32, vp]),
        "tbnav_rbpf_set_log_odds": (C.c_int, [vp, i32, vp]),
        "tbnav_rbpf_get_occ_dist": (C.c_int, [vp, i32, vp]),
        "tbnav_rbpf_set_occ_dist": (C.c_int, [vp, i32, vp]),
        "tbnav_rbpf_get_dist_code": (C.c_int, [vp, i32, vp]),
        "tbnav_rbpf_get_occupied_count": (C.c_int, [vp, vp]),
        "tbnav_rbpf_get_trace": (C.c_int, [vp] + [vp] * 9),
        "tbnav_rbpf_last_kernel_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
        "tbnav_rbpf_raycast_box_cells": (C.c_int, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "tbnav_rbpf_last_kernel_names": (C.c_int, [vp, C.c_char_p, i32, C.c_char_p, i32, C.POINTER(C.c_int32)]),
        "tbnav_rbpf_set_timing": (C.c_int, [vp, i32]),
        "tbnav_rbpf_set_scan_matching": (C.c_int, [vp, i32, C.c_double, C.c_double, i32]),
        "tbnav_rbpf_get_scan_match": (C.c_int, [vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(status: int, where: str) -> None:
    if status != OK:
        L = lib()
        detail = L.tbnav_status_string(status).decode()
        hip = L.tbnav_last_hip_error().decode()
        raise TbnavError(status, where, detail + (": " + hip if hip else ""))
