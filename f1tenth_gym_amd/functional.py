"""The reference's free kernel functions (star-exported from f110_gym.envs, envs/__init__.py:2-5)
with their original signatures, evaluated on the MI355X through the unit entry points of the
C ABI.  They exist for drop-in completeness and for the parity tests; batched callers should
use BatchSim.*_batch directly (one launch for M items instead of M launches)."""
import numpy as np

from .core import BatchSim, DEFAULT_PARAMS

_ctx = {}


def _unit_sim(num_beams=1080):
    if num_beams not in _ctx:
        _ctx[num_beams] = BatchSim(DEFAULT_PARAMS, num_envs=1, num_agents=1, num_beams=num_beams)
    return _ctx[num_beams]


def _pvec(mu, C_Sf, C_Sr, lf, lr, h, m, I, s_min, s_max, sv_min, sv_max, v_switch, a_max, v_min, v_max):
    return np.array([mu, C_Sf, C_Sr, lf, lr, h, m, I, s_min, s_max, sv_min, sv_max, v_switch, a_max,
                     v_min, v_max, 0.31, 0.58], dtype=np.float64)


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


def pid(speed, steer, current_speed, current_steer, max_sv, max_a, max_v, min_v):
    p = dict(DEFAULT_PARAMS)
    p.update({'sv_max': max_sv, 'a_max': max_a, 'v_max': max_v, 'v_min': min_v})
    out = _unit_sim().pid_batch(np.array([[speed, steer, current_speed, current_steer]], dtype=np.float64), p)
    return out[0, 0], out[0, 1]


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


def check_ttc_jit(scan, vel, scan_angles, cosines, side_distances, ttc_thresh):
    scan = np.asarray(scan, dtype=np.float64)
    b = BatchSim(DEFAULT_PARAMS, num_envs=1, num_agents=1, num_beams=scan.shape[0])
    try:
        from . import _ffi
        _ffi.check(_ffi.lib().f110_set_beam_tables(b._h, _ffi.dptr(_ffi.as_f64(scan_angles)),
                                                   _ffi.dptr(_ffi.as_f64(cosines)),
                                                   _ffi.dptr(_ffi.as_f64(side_distances)), scan.shape[0]), b._h)
        return bool(b.ttc_batch(scan.reshape(1, -1), np.array([vel], dtype=np.float64), ttc_thresh)[0])
    finally:
        b.close()


def ray_cast(pose, scan, scan_angles, vertices):
    """in-place on `scan` like the reference (laser_models.py:318-346)"""
    scan_angles = np.asarray(scan_angles, dtype=np.float64)
    b = BatchSim(DEFAULT_PARAMS, num_envs=1, num_agents=1, num_beams=scan_angles.shape[0])
    try:
        from . import _ffi
        _ffi.check(_ffi.lib().f110_set_beam_tables(b._h, _ffi.dptr(_ffi.as_f64(scan_angles)),
                                                   _ffi.dptr(_ffi.as_f64(np.cos(scan_angles))),
                                                   _ffi.dptr(_ffi.as_f64(np.ones_like(scan_angles))),
                                                   scan_angles.shape[0]), b._h)
        # the device kernel assumes uniformly spaced beam angles (base_classes.py:133-134)
        b_out, _ = b.raycast_batch(np.asarray(pose, dtype=np.float64).reshape(1, 3),
                                   np.asarray(vertices, dtype=np.float64).reshape(1, 4, 2),
                                   np.asarray(scan, dtype=np.float64).reshape(1, -1))
    finally:
        b.close()
    scan[:] = b_out[0]
    return scan
