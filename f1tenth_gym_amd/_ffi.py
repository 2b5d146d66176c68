"""ctypes binding of libf110_hip.so (include/f110.h).  No torch, no cffi.

The library is the product: if it cannot be loaded, or no MI355X is visible, every entry point
raises — there is deliberately no CPU fallback (the CPU oracle lives under oracle/ and is
test infrastructure only; nothing here imports it).
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# F110_LIB_VARIANT=experimental (read HERE, by the Python host — the library reads no environment) selects
# libf110_hip_exp.so: the product plus every variant that was measured and not adopted, and f110_exp_set
VARIANT = "experimental" if os.environ.get("F110_LIB_VARIANT", "product").lower().startswith("exp") else "product"
LIB_PATH = os.path.join(_PKG, "libf110_hip_exp.so" if VARIANT == "experimental" else "libf110_hip.so")
ABI_VERSION = 1
NPARAMS = 18
PARAM_KEYS = ['mu', 'C_Sf', 'C_Sr', 'lf', 'lr', 'h', 'm', 'I', 's_min', 's_max', 'sv_min',
              'sv_max', 'v_switch', 'a_max', 'v_min', 'v_max', 'width', 'length']

OK, ERR_INVALID, ERR_NO_MAP, ERR_HIP, ERR_STATE, ERR_NOMEM = 0, -1, -2, -3, -4, -5
MAP_ROWMAJOR_F64, MAP_TILED_F64, MAP_CODE8, MAP_PADDED_F64, MAP_WINDOW_LDS = 0, 1, 2, 3, 4
MAP_DEFAULT = MAP_PADDED_F64   # fastest; falls back to row-major for maps too large for it
INTEGRATOR_RK4, INTEGRATOR_EULER = 1, 2

_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("num_envs", C.c_int32), ("num_agents", C.c_int32),
                ("num_beams", C.c_int32), ("theta_dis", C.c_int32), ("integrator", C.c_int32),
                ("device_id", C.c_int32), ("map_layout", C.c_int32), ("scan_block", C.c_int32),
                ("scan_tasks_per_wave", C.c_int32), ("step_groups", C.c_int32), ("step_graph", C.c_int32), ("fov", C.c_double), ("eps", C.c_double),
                ("max_range", C.c_double), ("time_step", C.c_double), ("lidar_dist", C.c_double),
                ("ttc_thresh", C.c_double), ("params", C.c_double * NPARAMS)]


class ObsHost(C.Structure):
    _fields_ = [("scans", _dp), ("poses_x", _dp), ("poses_y", _dp), ("poses_theta", _dp),
                ("linear_vels_x", _dp), ("ang_vels_z", _dp), ("collisions", _dp),
                ("collision_idx", _dp), ("state", _dp), ("agent_poses", _dp),
                ("in_collision", _i32p), ("step_count", _i32p)]


class EpisodeHost(C.Structure):
    _fields_ = [("lap_times", _dp), ("lap_counts", _dp), ("toggles", _dp), ("current_time", _dp),
                ("near_starts", _u8p), ("done", _u8p), ("checkpoint_done", _u8p)]


class EpisodeViews(C.Structure):
    _fields_ = [("done", C.c_void_p), ("checkpoint_done", C.c_void_p), ("lap_times", C.c_void_p),
                ("lap_counts", C.c_void_p), ("toggles", C.c_void_p), ("current_time", C.c_void_p)]


class HostBlock(C.Structure):
    """f110_host_block: where f110_step_host writes (page-locked memory of f110_host_alloc)"""
    _fields_ = [("state", _dp), ("collisions", _dp), ("collision_idx", _dp), ("agent_poses", _dp),
                ("lap_times", _dp), ("lap_counts", _dp), ("toggles", _dp), ("current_time", _dp),
                ("in_collision", _i32p), ("near_starts", _u8p), ("checkpoint_done", _u8p), ("done", _u8p),
                ("scans", _dp)]


GATHER_F64, GATHER_F32 = 0, 1
STEP_AUTO_RESET, STEP_NO_SYNC, STEP_ACTIONS_MAPPED, STEP_SPIN_WAIT, STEP_NO_FUSE, STEP_POLL = 1, 2, 4, 8, 16, 32


class DeviceViews(C.Structure):
    _fields_ = [("scans", C.c_void_p), ("state", C.c_void_p), ("agent_poses", C.c_void_p),
                ("collisions", C.c_void_p), ("collision_idx", C.c_void_p),
                ("in_collision", C.c_void_p), ("step_count", C.c_void_p), ("stream", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/f110.h declares
PROTOTYPES = {
    "f110_last_error": (C.c_char_p, [C.c_void_p]),
    "f110_abi_version": (C.c_int, []),
    "f110_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "f110_device_pci_bus_id": (C.c_int, [C.c_int32, C.c_char_p, C.c_int32]),
    "f110_build_info": (C.c_char_p, []),
    "f110_is_experimental": (C.c_int, []),
    "f110_exp_set": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "f110_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "f110_destroy": (None, [C.c_void_p]),
    "f110_sync": (C.c_int, [C.c_void_p]),
    "f110_set_map_image": (C.c_int, [C.c_void_p, _u8p, C.c_int32, C.c_int32] + [C.c_double] * 4),
    "f110_set_map_dt": (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_int32] + [C.c_double] * 5),
    "f110_get_map_dt": (C.c_int, [C.c_void_p, _dp]),
    "f110_map_shape": (C.c_int, [C.c_void_p, _i32p, _i32p]),
    "f110_set_trig_tables": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int32]),
    "f110_set_beam_tables": (C.c_int, [C.c_void_p, _dp, _dp, _dp, C.c_int32]),
    "f110_set_params": (C.c_int, [C.c_void_p, C.c_int32, _dp]),
    "f110_set_noise_table": (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_int32]),
    "f110_set_noise_rng": (C.c_int, [C.c_void_p, _u64p, C.c_int32, C.c_double, C.c_int32]),
    "f110_pcg64_seed": (C.c_int, [C.c_uint64, _u64p]),
    "f110_noise_prepare": (C.c_int, [C.c_void_p, C.c_int32]),
    "f110_noise_rows_batch": (C.c_int, [C.c_void_p, _u64p, C.c_double, C.c_int32, C.c_int32, _dp, _u64p]),
    "f110_scan_lookup_count": (C.c_int, [C.c_void_p, C.c_int32, _i64p]),
    "f110_reset": (C.c_int, [C.c_void_p, _dp, _u8p]),
    "f110_reset_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "f110_add_map_image": (C.c_int, [C.c_void_p, _u8p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, _i32p]),
    "f110_add_map_dt": (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i32p]),
    "f110_set_env_maps": (C.c_int, [C.c_void_p, _i32p]),
    "f110_set_params_batch": (C.c_int, [C.c_void_p, _dp]),
    "f110_reset_collided_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "f110_set_auto_reseat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "f110_episode_init": (C.c_int, [C.c_void_p, C.c_int32]),
    "f110_episode_reset": (C.c_int, [C.c_void_p, _dp, _dp, _u8p]),
    "f110_episode_step_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "f110_episode_reset_done_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "f110_episode_get": (C.c_int, [C.c_void_p, C.POINTER(EpisodeHost)]),
    "f110_episode_step_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "f110_episode_packed_bytes": (C.c_size_t, [C.c_void_p]),
    "f110_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "f110_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "f110_episode_device_views": (C.c_int, [C.c_void_p, C.POINTER(EpisodeViews)]),
    "f110_step_host": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(HostBlock), C.c_int32]),
    "f110_step_host_stats": (C.c_int, [C.c_void_p, _dp]),
    "f110_step": (C.c_int, [C.c_void_p, _dp]),
    "f110_step_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "f110_get_obs": (C.c_int, [C.c_void_p, C.POINTER(ObsHost)]),
    "f110_set_state": (C.c_int, [C.c_void_p, _dp, _dp, _i32p]),
    "f110_get_device_views": (C.c_int, [C.c_void_p, C.POINTER(DeviceViews)]),
    "f110_stream_fence": (C.c_int, [C.c_void_p]),
    "f110_device_mem_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "f110_device_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "f110_device_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "f110_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "f110_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "f110_comm_unique_id": (C.c_int, [C.c_void_p]),
    "f110_comm_init": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "f110_comm_all_gather_scans": (C.c_int, [C.c_void_p, C.c_void_p]),
    "f110_comm_all_gather_obs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "f110_comm_gather_obs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "f110_comm_info": (C.c_int, [C.c_void_p, _i32p, _i32p]),
    "f110_step_groups": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p]),
    "f110_step_launches": (C.c_int, [C.c_void_p, _i32p]),
    "f110_comm_set_overlap": (C.c_int, [C.c_void_p, C.c_int32]),
    "f110_comm_destroy": (C.c_int, [C.c_void_p]),
    "f110_timer_begin": (C.c_int, [C.c_void_p]),
    "f110_timer_end_ms": (C.c_int, [C.c_void_p, _dp]),
    "f110_profile_kernels": (C.c_int, [C.c_void_p, C.c_int32]),
    "f110_profile_read": (C.c_int, [C.c_void_p, _i32p, _dp, _dp, _dp]),
    "f110_scan_batch": (C.c_int, [C.c_void_p, _dp, C.c_int32, _dp, _i32p, _i64p]),
    "f110_scan_path_stats": (C.c_int, [C.c_void_p, C.c_int32, _i64p]),
    "f110_pure_pursuit_batch": (C.c_int, [C.c_void_p, _dp, C.c_int32, _dp, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, _dp]),
    "f110_pure_pursuit_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "f110_scan_policy_device": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "f110_dynamics_batch": (C.c_int, [C.c_void_p, _dp, _dp, _dp, C.c_int32, _dp, _dp]),
    "f110_pid_batch": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int32, _dp]),
    "f110_update_pose_batch": (C.c_int, [C.c_void_p, _dp, _dp, _i32p, _dp, _dp, C.c_double, C.c_int32,
                                         C.c_double, C.c_int32, _dp, _dp, _i32p, _dp]),
    "f110_get_vertices_batch": (C.c_int, [C.c_void_p, _dp, C.c_double, C.c_double, C.c_int32, _dp]),
    "f110_gjk_batch": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int32, _i32p]),
    "f110_collision_multiple_batch": (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_int32, _dp, _dp]),
    "f110_ttc_batch": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int32, C.c_double, _i32p]),
    "f110_raycast_batch": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int32, _dp, _i32p]),
    "f110_get_range_batch": (C.c_int, [C.c_void_p, _dp, C.c_int32, _dp]),
    "f110_edt_sq": (C.c_int, [C.c_void_p, _u8p, C.c_int32, C.c_int32, _u32p]),
    "f110_beam_dir_index_batch": (C.c_int, [C.c_void_p, _dp, C.c_int32, _i32p]),
    "f110_dt_from_bitmap": (C.c_int, [C.c_void_p, _u8p, C.c_int32, C.c_int32, C.c_double, _dp]),
    "f110_helper_batch": (C.c_int, [C.c_void_p, C.c_int32, _dp, C.c_int32, C.c_int32, _dp]),
}

# f110_helper_batch ops (include/f110.h)
(OP_ACCL_CONSTRAINTS, OP_STEERING_CONSTRAINT, OP_CROSS, OP_ARE_COLLINEAR, OP_PERPENDICULAR, OP_TRIPLE_PRODUCT, OP_AVG_POINT,
 OP_FURTHEST_POINT, OP_SUPPORT, OP_GET_TRMTX, OP_XY_2_RC, OP_DISTANCE_TRANSFORM, OP_TRACE_RAY) = range(1, 14)

_lib = None


class F110LibraryError(RuntimeError):
    """libf110_hip.so is missing/unloadable, or the HIP runtime reported an error."""


class ExperimentalOnly(F110LibraryError):
    """the call needs libf110_hip_exp.so (F110_LIB_VARIANT=experimental): a layout, a step form or a
    switch that was measured and not adopted into the product library"""


def lib():
    """Load libf110_hip.so (built in-tree by f1tenth_gym_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise F110LibraryError(
                "%s not found — build it with `python -m f1tenth_gym_amd.build` "
                "(there is no CPU fallback for the MI355X hot path)" % LIB_PATH)
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as ex:
            raise F110LibraryError("cannot load %s: %s" % (LIB_PATH, ex))
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                raise F110LibraryError("%s does not export %s (stale build?)" % (LIB_PATH, name))
            fn.restype = res
            fn.argtypes = args
        if L.f110_abi_version() != ABI_VERSION:
            raise F110LibraryError("ABI version mismatch: library %d, binding %d"
                                   % (L.f110_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def last_error(handle=None):
    msg = lib().f110_last_error(handle)
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, handle=None, invalid_exc=ValueError):
    """Map a return code to the exception type the reference raises for the same condition."""
    if rc == OK:
        return
    msg = last_error(handle)
    if rc == ERR_NO_MAP:
        raise ValueError(msg or 'Map is not set for scan simulator.')   # laser_models.py:445-446
    if rc == ERR_INVALID:
        if "out of bounds for list of agents" in msg:
            raise IndexError(msg)                                        # base_classes.py:534
        if "Invalid Integrator" in msg:
            raise SyntaxError(msg)                                       # base_classes.py:398
        raise invalid_exc(msg)
    if rc == ERR_NOMEM:
        raise MemoryError(msg)
    if rc == ERR_STATE and "experimental build only" in msg:
        raise ExperimentalOnly(msg)
    raise F110LibraryError(msg or ("libf110_hip error %d" % rc))


def device_count():
    n = C.c_int(0)
    rc = lib().f110_device_count(C.byref(n))
    return n.value if rc == OK else 0


def as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("expected array of shape %s, got %s" % (tuple(shape), tuple(a.shape)))
    return a


def dptr(a):
    return a.ctypes.data_as(_dp)


def i32ptr(a):
    return a.ctypes.data_as(_i32p)


def params_vector(params):
    try:
        return np.array([float(params[k]) for k in PARAM_KEYS], dtype=np.float64)
    except KeyError as ex:
        raise KeyError("vehicle params dict is missing %s" % ex)
