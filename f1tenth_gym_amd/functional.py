"""The reference's free functions (star-exported from f110_gym.envs, envs/__init__.py:2-5) with their original
signatures, evaluated on the MI355X through the unit entry points of the C ABI (include/f110.h).  They exist for drop-in
completeness and as addressable parity targets (tests/test_gpu_round6.py holds every one of them against reference-generated
rows); batched callers should use BatchSim.*_batch directly (one launch for M items instead of M launches).

Handles are cached: one per beam count for the functions that take beam tables (their tables are re-uploaded only when the
caller's arrays change), one per distance table for the functions that take `dt` (keyed by the array's address, shape and an
Adler-32 of its bytes, so a table edited in place is noticed).  `close_cached_handles()` releases them.

`scan_angles` may be ANY array: a table that is not the uniform ramp RaceCar builds (base_classes.py:133-134) takes the
reference's full first-minimum argmin per vertex (laser_models.py:310-313) in the unit kernel."""
import zlib

import numpy as np

from . import _ffi
from .core import BatchSim, DEFAULT_PARAMS

_ctx = {}        # num_beams -> BatchSim for functions that need no map
_maps = {}       # dt key -> BatchSim holding that distance table
_MAX_MAP_HANDLES = 4


def _unit_sim(num_beams=1080):
    if num_beams not in _ctx:
        _ctx[num_beams] = BatchSim(DEFAULT_PARAMS, num_envs=1, num_agents=1, num_beams=int(num_beams))
        _ctx[num_beams]._tables = None
    return _ctx[num_beams]


def _beam_sim(scan_angles, cosines=None, side_distances=None):
    """the cached handle of this beam count with these per-beam tables on the device.  cosines / side_distances None (ray_cast,
    get_blocked_view_indices: only the angles matter): whatever the handle holds for them stays, so a loop that alternates
    check_ttc_jit and ray_cast with one scan_angles table uploads nothing after its first two calls"""
    sa = _ffi.as_f64(scan_angles).reshape(-1)
    b = _unit_sim(sa.shape[0])
    t = b._tables
    co = _ffi.as_f64(cosines, sa.shape) if cosines is not None else (t[1] if t is not None else np.cos(sa))
    sd = _ffi.as_f64(side_distances, sa.shape) if side_distances is not None else (t[2] if t is not None else np.ones_like(sa))
    if t is None or not (np.array_equal(t[0], sa, equal_nan=True) and np.array_equal(t[1], co, equal_nan=True)
                         and np.array_equal(t[2], sd, equal_nan=True)):
        b.set_beam_tables(sa, co, sd)
        b._tables = (sa.copy(), co.copy(), sd.copy())
    return b


def _map_sim(dt, resolution, orig_x, orig_y, orig_c, orig_s, num_beams=1080, fov=4.7, eps=0.0001, theta_dis=2000, max_range=30.0,
             sines=None, cosines=None):
    """the cached handle that holds distance table `dt` (and, for the scan functions, this trig table / these scan constants)"""
    dt = np.ascontiguousarray(dt, dtype=np.float64)
    key = (dt.ctypes.data, dt.shape, float(resolution), float(orig_x), float(orig_y), float(orig_c), float(orig_s), int(num_beams), float(fov),
           float(eps), int(theta_dis), float(max_range))
    ent = _maps.get(key)
    sums = (zlib.adler32(memoryview(dt).cast("B")),
            None if sines is None else zlib.adler32(np.ascontiguousarray(sines, dtype=np.float64).tobytes() + np.ascontiguousarray(cosines, dtype=np.float64).tobytes()))
    if ent is None or ent[1] != sums:
        if ent is not None:
            ent[0].close()
        elif len(_maps) >= _MAX_MAP_HANDLES:
            _maps.pop(next(iter(_maps)))[0].close()
        b = BatchSim(DEFAULT_PARAMS, num_envs=1, num_agents=1, num_beams=int(num_beams), fov=float(fov), eps=float(eps), theta_dis=int(theta_dis),
                     max_range=float(max_range))
        if sines is not None:
            s, c = _ffi.as_f64(sines, (int(theta_dis),)), _ffi.as_f64(cosines, (int(theta_dis),))
            _ffi.check(_ffi.lib().f110_set_trig_tables(b._h, _ffi.dptr(s), _ffi.dptr(c), int(theta_dis)), b._h)
        _ffi.check(_ffi.lib().f110_set_map_dt(b._h, _ffi.dptr(dt), dt.shape[0], dt.shape[1], float(resolution), float(orig_x), float(orig_y),
                                              float(orig_c), float(orig_s)), b._h)
        b.has_map = True
        _maps[key] = ent = (b, sums)
    return ent[0]


def close_cached_handles():
    for b in list(_ctx.values()) + [e[0] for e in _maps.values()]:
        b.close()
    _ctx.clear()
    _maps.clear()


def _pvec(mu, C_Sf, C_Sr, lf, lr, h, m, I, s_min, s_max, sv_min, sv_max, v_switch, a_max, v_min, v_max):
    return np.array([mu, C_Sf, C_Sr, lf, lr, h, m, I, s_min, s_max, sv_min, sv_max, v_switch, a_max,
                     v_min, v_max, 0.31, 0.58], dtype=np.float64)


def _row(*vals):
    return np.array([[float(v) for v in vals]], dtype=np.float64)


# ------------------------------------------------------------------ dynamic_models.py
def accl_constraints(vel, accl, v_switch, a_max, v_min, v_max):
    """dynamic_models.py:29-60"""
    return float(_unit_sim().helper_batch(_ffi.OP_ACCL_CONSTRAINTS, _row(vel, accl, v_switch, a_max, v_min, v_max))[0, 0])


def steering_constraint(steering_angle, steering_velocity, s_min, s_max, sv_min, sv_max):
    """dynamic_models.py:62-87"""
    return float(_unit_sim().helper_batch(_ffi.OP_STEERING_CONSTRAINT, _row(steering_angle, steering_velocity, s_min, s_max, sv_min, sv_max))[0, 0])


def vehicle_dynamics_st(x, u_init, mu, C_Sf, C_Sr, lf, lr, h, m, I, s_min, s_max, sv_min, sv_max, v_switch,
                        a_max, v_min, v_max):
    f_st, _ = _unit_sim().dynamics_batch(np.asarray(x, dtype=np.float64).reshape(1, 7),
                                         np.asarray(u_init, dtype=np.float64).reshape(1, 2),
                                         _pvec(mu, C_Sf, C_Sr, lf, lr, h, m, I, s_min, s_max, sv_min, sv_max,
                                               v_switch, a_max, v_min, v_max))
    return f_st[0]


def vehicle_dynamics_ks(x, u_init, mu, C_Sf, C_Sr, lf, lr, h, m, I, s_min, s_max, sv_min, sv_max, v_switch,
                        a_max, v_min, v_max):
    x7 = np.zeros((1, 7))
    x7[0, :5] = np.asarray(x, dtype=np.float64)[:5]
    _, f_ks = _unit_sim().dynamics_batch(x7, np.asarray(u_init, dtype=np.float64).reshape(1, 2),
                                         _pvec(mu, C_Sf, C_Sr, lf, lr, h, m, I, s_min, s_max, sv_min, sv_max,
                                               v_switch, a_max, v_min, v_max))
    return f_ks[0]


def func_KS(x, t, u, *params):
    """dynamic_models.py:223-225 (the odeint right-hand side of the reference's own tests)"""
    return vehicle_dynamics_ks(x, u, *params)


def func_ST(x, t, u, *params):
    """dynamic_models.py:227-229"""
    return vehicle_dynamics_st(x, u, *params)


def pid(speed, steer, current_speed, current_steer, max_sv, max_a, max_v, min_v):
    p = dict(DEFAULT_PARAMS)
    p.update({'sv_max': max_sv, 'a_max': max_a, 'v_max': max_v, 'v_min': min_v})
    out = _unit_sim().pid_batch(np.array([[speed, steer, current_speed, current_steer]], dtype=np.float64), p)
    return out[0, 0], out[0, 1]


# ------------------------------------------------------------------ collision_models.py
def perpendicular(pt):
    """collision_models.py:34-48 — in place on `pt`, like the reference"""
    out = _unit_sim().helper_batch(_ffi.OP_PERPENDICULAR, np.asarray(pt, dtype=np.float64).reshape(1, 2))[0]
    pt[0], pt[1] = out[0], out[1]
    return pt


def tripleProduct(a, b, c):
    """collision_models.py:51-64"""
    return _unit_sim().helper_batch(_ffi.OP_TRIPLE_PRODUCT, np.concatenate([np.asarray(v, dtype=np.float64).reshape(2) for v in (a, b, c)]).reshape(1, 6))[0]


def avgPoint(vertices):
    """collision_models.py:67-78"""
    v = np.asarray(vertices, dtype=np.float64)
    return _unit_sim().helper_batch(_ffi.OP_AVG_POINT, v.reshape(1, -1), n=v.shape[0])[0]


def indexOfFurthestPoint(vertices, d):
    """collision_models.py:81-92"""
    v = np.asarray(vertices, dtype=np.float64)
    row = np.concatenate([v.reshape(-1), np.asarray(d, dtype=np.float64).reshape(2)]).reshape(1, -1)
    return int(_unit_sim().helper_batch(_ffi.OP_FURTHEST_POINT, row, n=v.shape[0])[0, 0])


def support(vertices1, vertices2, d):
    """collision_models.py:95-110"""
    v1, v2 = np.asarray(vertices1, dtype=np.float64), np.asarray(vertices2, dtype=np.float64)
    if v1.shape != v2.shape:
        raise ValueError("support: the two bodies must have the same number of vertices on this path")
    row = np.concatenate([v1.reshape(-1), v2.reshape(-1), np.asarray(d, dtype=np.float64).reshape(2)]).reshape(1, -1)
    return _unit_sim().helper_batch(_ffi.OP_SUPPORT, row, n=v1.shape[0])[0]


def get_trmtx(pose):
    """collision_models.py:218-235"""
    return _unit_sim().helper_batch(_ffi.OP_GET_TRMTX, np.asarray(pose, dtype=np.float64).reshape(1, 3))[0].reshape(4, 4)


def get_vertices(pose, length, width):
    return _unit_sim().get_vertices_batch(np.asarray(pose, dtype=np.float64).reshape(1, 3), length, width)[0]


def collision(vertices1, vertices2):
    va = np.asarray(vertices1, dtype=np.float64).reshape(1, 4, 2)
    vb = np.asarray(vertices2, dtype=np.float64).reshape(1, 4, 2)
    return bool(_unit_sim().gjk_batch(va, vb)[0])


def collision_multiple(vertices):
    v = np.asarray(vertices, dtype=np.float64)
    col, idx = _unit_sim().collision_multiple_batch(v.reshape(1, v.shape[0], 4, 2))
    return col[0], idx[0]


# ------------------------------------------------------------------ laser_models.py
def get_dt(bitmap, resolution):
    """laser_models.py:40-53: resolution * distance_transform_edt(bitmap), the exact EDT on the device"""
    return _unit_sim().dt_from_bitmap(bitmap, resolution)


def xy_2_rc(x, y, orig_x, orig_y, orig_c, orig_s, height, width, resolution):
    """laser_models.py:55-86 -> (r, c)"""
    r, c = _unit_sim().helper_batch(_ffi.OP_XY_2_RC, _row(x, y, orig_x, orig_y, orig_c, orig_s, height, width, resolution))[0]
    return int(r), int(c)


def _check_shape(dt, height, width):
    if np.shape(dt) != (int(height), int(width)):
        raise ValueError("dt has shape %s, height x width is %d x %d" % (np.shape(dt), int(height), int(width)))


def distance_transform(x, y, orig_x, orig_y, orig_c, orig_s, height, width, resolution, dt):
    """laser_models.py:88-104"""
    _check_shape(dt, height, width)
    b = _map_sim(dt, resolution, orig_x, orig_y, orig_c, orig_s)
    return float(b.helper_batch(_ffi.OP_DISTANCE_TRANSFORM, _row(x, y))[0, 0])


def trace_ray(x, y, theta_index, sines, cosines, eps, orig_x, orig_y, orig_c, orig_s, height, width, resolution, dt, max_range):
    """laser_models.py:106-146"""
    _check_shape(dt, height, width)
    b = _map_sim(dt, resolution, orig_x, orig_y, orig_c, orig_s, eps=eps, theta_dis=len(sines), max_range=max_range, sines=sines, cosines=cosines)
    return float(b.helper_batch(_ffi.OP_TRACE_RAY, _row(x, y, theta_index))[0, 0])


def get_scan(pose, theta_dis, fov, num_beams, theta_index_increment, sines, cosines, eps, orig_x, orig_y, orig_c, orig_s, height, width,
             resolution, dt, max_range):
    """laser_models.py:148-186.  theta_index_increment must be the value ScanSimulator2D derives from the other arguments
    (theta_dis * (fov / (num_beams - 1)) / (2 pi), :367-368) — the only one the reference itself ever passes; anything else raises."""
    _check_shape(dt, height, width)
    derived = theta_dis * (float(fov) / (int(num_beams) - 1)) / (2. * np.pi)
    if float(theta_index_increment) != derived:
        raise ValueError("get_scan: theta_index_increment %r is not theta_dis*fov/(num_beams-1)/(2 pi) = %r; the device kernel derives the "
                         "increment from (theta_dis, fov, num_beams) like ScanSimulator2D does" % (float(theta_index_increment), derived))
    b = _map_sim(dt, resolution, orig_x, orig_y, orig_c, orig_s, num_beams=num_beams, fov=fov, eps=eps, theta_dis=theta_dis, max_range=max_range,
                 sines=sines, cosines=cosines)
    return b.scan_batch(np.asarray(pose, dtype=np.float64).reshape(1, 3))[0]


def check_ttc_jit(scan, vel, scan_angles, cosines, side_distances, ttc_thresh):
    """laser_models.py:188-217"""
    scan = np.asarray(scan, dtype=np.float64)
    b = _beam_sim(scan_angles, cosines, side_distances)
    return bool(b.ttc_batch(scan.reshape(1, -1), np.array([vel], dtype=np.float64), ttc_thresh)[0])


def cross(v1, v2):
    """laser_models.py:219-230"""
    return float(_unit_sim().helper_batch(_ffi.OP_CROSS, _row(v1[0], v1[1], v2[0], v2[1]))[0, 0])


def are_collinear(pt_a, pt_b, pt_c):
    """laser_models.py:232-247"""
    return bool(_unit_sim().helper_batch(_ffi.OP_ARE_COLLINEAR, _row(pt_a[0], pt_a[1], pt_b[0], pt_b[1], pt_c[0], pt_c[1]))[0, 0])


def get_range(pose, beam_theta, va, vb):
    """laser_models.py:249-280"""
    return float(_unit_sim().get_range_batch(_row(pose[0], pose[1], pose[2], beam_theta, va[0], va[1], vb[0], vb[1]))[0])


def get_blocked_view_indices(pose, vertices, scan_angles):
    """laser_models.py:282-315 -> (min_ind, max_ind)"""
    b = _beam_sim(scan_angles, None, None)
    dummy = np.zeros((1, b.B))    # ranges of 0: no beam is lowered, only the window is wanted
    _, mm = b.raycast_batch(np.asarray(pose, dtype=np.float64).reshape(1, 3), np.asarray(vertices, dtype=np.float64).reshape(1, 4, 2), dummy)
    return int(mm[0, 0]), int(mm[0, 1])


def ray_cast(pose, scan, scan_angles, vertices):
    """laser_models.py:318-346 — in place on `scan`, like the reference"""
    b = _beam_sim(scan_angles, None, None)
    b_out, _ = b.raycast_batch(np.asarray(pose, dtype=np.float64).reshape(1, 3),
                               np.asarray(vertices, dtype=np.float64).reshape(1, 4, 2),
                               np.asarray(scan, dtype=np.float64).reshape(1, -1))
    scan[:] = b_out[0]
    return scan
