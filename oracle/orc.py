"""ctypes binding of oracle/libf110_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product package f1tenth_gym_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libf110_oracle.so")
_lib = None

PARAM_KEYS = ['mu', 'C_Sf', 'C_Sr', 'lf', 'lr', 'h', 'm', 'I', 's_min', 's_max', 'sv_min',
              'sv_max', 'v_switch', 'a_max', 'v_min', 'v_max', 'width', 'length']
DEFAULT_PARAMS = {'mu': 1.0489, 'C_Sf': 4.718, 'C_Sr': 5.4562, 'lf': 0.15875, 'lr': 0.17145,
                  'h': 0.074, 'm': 3.74, 'I': 0.04712, 's_min': -0.4189, 's_max': 0.4189,
                  'sv_min': -3.2, 'sv_max': 3.2, 'v_switch': 7.319, 'a_max': 9.51,
                  'v_min': -5.0, 'v_max': 20.0, 'width': 0.31, 'length': 0.58}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_i64p = C.POINTER(C.c_int64)


class ScanCfg(C.Structure):
    _fields_ = [("num_beams", C.c_int), ("fov", C.c_double), ("eps", C.c_double),
                ("max_range", C.c_double), ("theta_dis", C.c_int),
                ("angle_increment", C.c_double), ("theta_index_increment", C.c_double),
                ("sines", _dp), ("cosines", _dp), ("height", C.c_int), ("width", C.c_int),
                ("resolution", C.c_double), ("orig_x", C.c_double), ("orig_y", C.c_double),
                ("orig_c", C.c_double), ("orig_s", C.c_double), ("dt", _dp)]


def build(force=False):
    if force or not os.path.isfile(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "f110_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
    return _LIB_PATH


def _bind(L):
    """restype / argtypes of everything oracle/f110_oracle.h declares"""
    L.orc_accl_constraints.restype = C.c_double
    L.orc_accl_constraints.argtypes = [C.c_double] * 6
    L.orc_steering_constraint.restype = C.c_double
    L.orc_steering_constraint.argtypes = [C.c_double] * 6
    L.orc_vehicle_dynamics_ks.argtypes = [_dp, _dp, _dp, _dp]
    L.orc_vehicle_dynamics_st.argtypes = [_dp, _dp, _dp, _dp]
    L.orc_pid.argtypes = [C.c_double] * 8 + [_dp, _dp]
    L.orc_update_pose.argtypes = [_dp, _dp, _ip, C.c_double, C.c_double, _dp, C.c_double,
                                  C.c_int, C.c_double, _dp]
    L.orc_xy_2_rc.argtypes = [C.POINTER(ScanCfg), C.c_double, C.c_double, _ip, _ip]
    L.orc_trace_ray.restype = C.c_double
    L.orc_trace_ray.argtypes = [C.POINTER(ScanCfg), C.c_double, C.c_double, C.c_double, _ip, _i64p]
    L.orc_get_scan.argtypes = [C.POINTER(ScanCfg), _dp, _dp, _ip, _i64p]
    L.orc_beam_dir_indices.argtypes = [C.POINTER(ScanCfg), C.c_double, _ip]
    L.orc_check_ttc.restype = C.c_int
    L.orc_check_ttc.argtypes = [_dp, C.c_int, C.c_double, _dp, _dp, C.c_double]
    L.orc_get_range.restype = C.c_double
    L.orc_get_range.argtypes = [_dp, C.c_double, _dp, _dp]
    L.orc_get_blocked_view_indices.argtypes = [_dp, _dp, _dp, C.c_int, _ip, _ip]
    L.orc_ray_cast.argtypes = [_dp, _dp, _dp, C.c_int, _dp]
    L.orc_build_beam_tables.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double,
                                        C.c_double, _dp, _dp, _dp]
    L.orc_edt_sq.argtypes = [_u8p, C.c_int, C.c_int, _u32p]
    L.orc_map_dt_from_image.argtypes = [_u8p, C.c_int, C.c_int, C.c_double, _dp]
    L.orc_get_vertices.argtypes = [_dp, C.c_double, C.c_double, _dp]
    L.orc_collision.restype = C.c_int
    L.orc_collision.argtypes = [_dp, _dp]
    L.orc_collision_multiple.argtypes = [_dp, C.c_int, _dp, _dp]
    L.orc_sim_create.restype = C.c_void_p
    L.orc_sim_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                 C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, _dp]
    L.orc_sim_destroy.argtypes = [C.c_void_p]
    L.orc_sim_set_tables.argtypes = [C.c_void_p, _dp, _dp]
    L.orc_sim_set_map_dt.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int] + [C.c_double] * 5
    L.orc_sim_set_params.restype = C.c_int
    L.orc_sim_set_params.argtypes = [C.c_void_p, C.c_int, _dp]
    L.orc_sim_set_noise.argtypes = [C.c_void_p, _dp, C.c_int]
    L.orc_sim_reset.argtypes = [C.c_void_p, _dp, _u8p]
    L.orc_sim_step.argtypes = [C.c_void_p, _dp, C.c_int]
    for name, ty in [("state", _dp), ("scans", _dp), ("collisions", _dp),
                     ("collision_idx", _dp), ("agent_poses", _dp), ("in_collision", _i32p),
                     ("step_count", _i32p), ("hit_rc", _i32p)]:
        fn = getattr(L, "orc_sim_" + name)
        fn.restype = ty
        fn.argtypes = [C.c_void_p]
    L.orc_sim_lookups.restype = C.c_int64
    L.orc_sim_lookups.argtypes = [C.c_void_p]
    L.orc_nearest_on_trajectory.restype = C.c_int
    L.orc_nearest_on_trajectory.argtypes = [_dp, C.c_int, C.c_double, C.c_double, _dp, _dp]
    L.orc_first_point_on_circle.restype = C.c_int
    L.orc_first_point_on_circle.argtypes = [_dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
    L.orc_pure_pursuit_plan.restype = None
    L.orc_pure_pursuit_plan.argtypes = [_dp, C.c_int, _dp, C.c_double, C.c_double, C.c_double, C.c_double, _dp]
    L.orc_sim_rollout.restype = C.c_int64
    L.orc_sim_rollout.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_int]
    return L


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _bind(C.CDLL(_LIB_PATH))
    return _lib


_NATIVE_DIR = os.path.join(_HERE, "_native")
_native = None
NATIVE_CFLAGS = ["-O3", "-march=native", "-fPIC", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fno-builtin", "-fopenmp", "-D_GNU_SOURCE"]


def native_lib():
    """bench.py's cpu_baseline only: the same source compiled FOR THE MACHINE IT RUNS ON (-O3 -march=native; the strict
    float64 flags of the Makefile stay), built where it is used — the checker library above is built once, portably,
    and travels to the GPU box."""
    global _native
    if _native is None:
        os.makedirs(_NATIVE_DIR, exist_ok=True)
        out = os.path.join(_NATIVE_DIR, "libf110_oracle_native.so")
        subprocess.check_call([os.environ.get("CC", "gcc")] + NATIVE_CFLAGS + ["-shared", "-o", out, os.path.join(_HERE, "f110_oracle.c"), "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _native = _bind(C.CDLL(out))
    return _native


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def params_vec(params=None):
    p = dict(DEFAULT_PARAMS)
    if params:
        p.update(params)
    return np.array([p[k] for k in PARAM_KEYS], dtype=np.float64)


# ---------------------------------------------------------------- function-level wrappers

def vehicle_dynamics_st(x, u, pvec):
    x, xp = _d(x); u, up = _d(u); pvec, pp = _d(pvec)
    f = np.empty(7); lib().orc_vehicle_dynamics_st(xp, up, pp, f.ctypes.data_as(_dp))
    return f


def vehicle_dynamics_ks(x, u, pvec):
    x, xp = _d(x); u, up = _d(u); pvec, pp = _d(pvec)
    f = np.empty(5); lib().orc_vehicle_dynamics_ks(xp, up, pp, f.ctypes.data_as(_dp))
    return f


def pid(speed, steer, current_speed, current_steer, max_sv, max_a, max_v, min_v):
    a = C.c_double(); s = C.c_double()
    lib().orc_pid(speed, steer, current_speed, current_steer, max_sv, max_a, max_v, min_v,
                  C.byref(a), C.byref(s))
    return a.value, s.value


def update_pose(state, steer_buf, buf_count, raw_steer, vel, pvec, time_step, integrator,
                lidar_dist):
    """returns (state, steer_buf, buf_count, scan_pose) — inputs are not modified."""
    st = np.array(state, dtype=np.float64); sb = np.array(steer_buf, dtype=np.float64)
    pvec, pp = _d(pvec)
    bc = C.c_int(int(buf_count)); sp = np.empty(3)
    lib().orc_update_pose(st.ctypes.data_as(_dp), sb.ctypes.data_as(_dp), C.byref(bc),
                          float(raw_steer), float(vel), pp, float(time_step), int(integrator),
                          float(lidar_dist), sp.ctypes.data_as(_dp))
    return st, sb, bc.value, sp


class ScanOracle(object):
    """ScanSimulator2D restated (laser_models.py:348-457), noise-free."""

    def __init__(self, num_beams, fov, eps=0.0001, theta_dis=2000, max_range=30.0):
        self.cfg = ScanCfg()
        c = self.cfg
        c.num_beams = num_beams; c.fov = fov; c.eps = eps; c.theta_dis = theta_dis
        c.max_range = max_range
        c.angle_increment = fov / (num_beams - 1)
        c.theta_index_increment = theta_dis * c.angle_increment / (2. * np.pi)
        theta_arr = np.linspace(0.0, 2 * np.pi, num=theta_dis)
        self.sines = np.sin(theta_arr); self.cosines = np.cos(theta_arr)
        c.sines = self.sines.ctypes.data_as(_dp); c.cosines = self.cosines.ctypes.data_as(_dp)
        self.dt = None

    def set_map_dt(self, dt, resolution, origin):
        self.dt = np.ascontiguousarray(dt, dtype=np.float64)
        c = self.cfg
        c.height, c.width = self.dt.shape
        c.resolution = resolution
        c.orig_x = origin[0]; c.orig_y = origin[1]
        c.orig_s = np.sin(origin[2]); c.orig_c = np.cos(origin[2])
        c.dt = self.dt.ctypes.data_as(_dp)

    def scan(self, pose, want_hits=False):
        pose, pp = _d(pose)
        B = self.cfg.num_beams
        out = np.empty(B); hits = np.empty((B, 2), dtype=np.intc)
        n = C.c_int64(0)
        lib().orc_get_scan(C.byref(self.cfg), pp, out.ctypes.data_as(_dp),
                           hits.ctypes.data_as(_ip), C.byref(n))
        self.last_lookups = n.value
        return (out, hits) if want_hits else out

    def beam_dir_indices(self, theta):
        idx = np.empty(self.cfg.num_beams, dtype=np.intc)
        lib().orc_beam_dir_indices(C.byref(self.cfg), float(theta), idx.ctypes.data_as(_ip))
        return idx

    def xy_2_rc(self, x, y):
        r = C.c_int(); c = C.c_int()
        lib().orc_xy_2_rc(C.byref(self.cfg), float(x), float(y), C.byref(r), C.byref(c))
        return r.value, c.value


def build_beam_tables(num_beams, fov, width, lf, lr):
    sa = np.empty(num_beams); co = np.empty(num_beams); sd = np.empty(num_beams)
    lib().orc_build_beam_tables(num_beams, fov, width, lf, lr, sa.ctypes.data_as(_dp),
                                co.ctypes.data_as(_dp), sd.ctypes.data_as(_dp))
    return sa, co, sd


def check_ttc(scan, vel, cosines, side_distances, ttc_thresh):
    scan, sp = _d(scan); cosines, cp = _d(cosines); side_distances, dp_ = _d(side_distances)
    return bool(lib().orc_check_ttc(sp, scan.shape[0], float(vel), cp, dp_, float(ttc_thresh)))


def get_range(pose, beam_theta, va, vb):
    pose, pp = _d(pose); va, ap = _d(va); vb, bp = _d(vb)
    return lib().orc_get_range(pp, float(beam_theta), ap, bp)


def get_blocked_view_indices(pose, vertices, scan_angles):
    pose, pp = _d(pose); vertices, vp = _d(vertices); scan_angles, sp = _d(scan_angles)
    lo = C.c_int(); hi = C.c_int()
    lib().orc_get_blocked_view_indices(pp, vp, sp, scan_angles.shape[0], C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def ray_cast(pose, scan, scan_angles, vertices):
    pose, pp = _d(pose); vertices, vp = _d(vertices); scan_angles, sp = _d(scan_angles)
    out = np.array(scan, dtype=np.float64)
    lib().orc_ray_cast(pp, out.ctypes.data_as(_dp), sp, out.shape[0], vp)
    return out


def edt_sq(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty(img.shape, dtype=np.uint32)
    lib().orc_edt_sq(img.ctypes.data_as(_u8p), img.shape[0], img.shape[1], out.ctypes.data_as(_u32p))
    return out


def map_dt_from_image(img_top_first, resolution):
    img = np.ascontiguousarray(img_top_first, dtype=np.uint8)
    out = np.empty(img.shape, dtype=np.float64)
    lib().orc_map_dt_from_image(img.ctypes.data_as(_u8p), img.shape[0], img.shape[1],
                                float(resolution), out.ctypes.data_as(_dp))
    return out


def get_vertices(pose, length, width):
    pose, pp = _d(pose)
    v = np.empty((4, 2)); lib().orc_get_vertices(pp, float(length), float(width), v.ctypes.data_as(_dp))
    return v


def collision(v1, v2):
    v1, p1 = _d(v1); v2, p2 = _d(v2)
    return bool(lib().orc_collision(p1, p2))


def collision_multiple(vertices):
    vertices, vp = _d(vertices)
    n = vertices.shape[0]
    col = np.empty(n); idx = np.empty(n)
    lib().orc_collision_multiple(vp, n, col.ctypes.data_as(_dp), idx.ctypes.data_as(_dp))
    return col, idx


class SimOracle(object):
    """Batched Simulator (base_classes.py:451-630): num_envs independent envs x num_agents."""

    def __init__(self, num_envs, num_agents, params=None, num_beams=1080, fov=4.7, eps=1e-4,
                 theta_dis=2000, max_range=30.0, time_step=0.01, integrator=1, lidar_dist=0.0,
                 ttc_thresh=0.005, native=False):
        self.E, self.A, self.N, self.B = num_envs, num_agents, num_envs * num_agents, num_beams
        self._L = native_lib() if native else lib()
        pv, pp = _d(params_vec(params))
        self._h = self._L.orc_sim_create(num_envs, num_agents, num_beams, fov, eps, theta_dis,
                                       max_range, time_step, integrator, lidar_dist, ttc_thresh, pp)
        theta_arr = np.linspace(0.0, 2 * np.pi, num=theta_dis)
        s, sp = _d(np.sin(theta_arr)); c, cp = _d(np.cos(theta_arr))
        self._L.orc_sim_set_tables(self._h, sp, cp)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_sim_destroy(self._h)
            self._h = None

    def set_map_dt(self, dt, resolution, origin):
        dt, p = _d(dt)
        self._L.orc_sim_set_map_dt(self._h, p, dt.shape[0], dt.shape[1], float(resolution),
                                 float(origin[0]), float(origin[1]), float(np.cos(origin[2])),
                                 float(np.sin(origin[2])))

    def set_params(self, params, agent_idx=-1):
        pv, pp = _d(params_vec(params))
        if self._L.orc_sim_set_params(self._h, agent_idx, pp) != 0:
            raise IndexError('Index given is out of bounds for list of agents.')

    def set_noise(self, noise):
        if noise is None:
            self._L.orc_sim_set_noise(self._h, None, 0)
        else:
            noise, p = _d(noise)
            self._L.orc_sim_set_noise(self._h, p, noise.shape[0])

    def reset(self, poses, env_mask=None):
        poses, p = _d(poses)
        assert poses.shape == (self.N, 3)
        if env_mask is None:
            self._L.orc_sim_reset(self._h, p, None)
        else:
            m = np.ascontiguousarray(env_mask, dtype=np.uint8)
            self._L.orc_sim_reset(self._h, p, m.ctypes.data_as(_u8p))

    def step(self, actions, n_threads=1):
        actions, p = _d(actions)
        assert actions.shape == (self.N, 2)
        self._L.orc_sim_step(self._h, p, int(n_threads))

    def rollout(self, action_sets, steps, steps_per_set, start_poses, reseat=True, n_threads=1):
        """every env through `steps` steps on its own (orc_sim_rollout); action_sets [n_sets][N][2]; -> re-seats"""
        acts, ap = _d(action_sets)
        assert acts.ndim == 3 and acts.shape[1:] == (self.N, 2)
        poses, pp = _d(start_poses)
        assert poses.shape == (self.N, 3)
        return int(self._L.orc_sim_rollout(self._h, ap, acts.shape[0], int(steps_per_set), int(steps), pp, 1 if reseat else 0, int(n_threads)))

    def _view(self, name, shape, dtype=np.float64):
        ptr = getattr(self._L, "orc_sim_" + name)(self._h)
        return np.ctypeslib.as_array(ptr, shape=shape)

    @property
    def state(self): return self._view("state", (self.N, 7))
    @property
    def scans(self): return self._view("scans", (self.N, self.B))
    @property
    def collisions(self): return self._view("collisions", (self.N,))
    @property
    def collision_idx(self): return self._view("collision_idx", (self.N,))
    @property
    def agent_poses(self): return self._view("agent_poses", (self.N, 3))
    @property
    def in_collision(self): return self._view("in_collision", (self.N,))
    @property
    def step_count(self): return self._view("step_count", (self.N,))
    @property
    def hit_rc(self): return self._view("hit_rc", (self.N, self.B, 2))
    @property
    def lookups(self): return self._L.orc_sim_lookups(self._h)


# ---- examples/waypoint_follow.py (pure-pursuit planner) ----
def nearest_on_trajectory(waypoints, px, py):
    w, wp = _d(waypoints)
    dist = C.c_double(0.0); t = C.c_double(0.0)
    i = lib().orc_nearest_on_trajectory(wp, w.shape[0], px, py, C.byref(dist), C.byref(t))
    return i, dist.value, t.value


def first_point_on_circle(waypoints, px, py, radius, start):
    w, wp = _d(waypoints)
    return lib().orc_first_point_on_circle(wp, w.shape[0], px, py, radius, start)


def pure_pursuit_plan(waypoints, pose, lookahead, vgain, wheelbase, max_reacquire=20.0):
    """-> (steer, speed)"""
    w, wp = _d(waypoints); p, pp = _d(pose)
    out = np.empty(2)
    lib().orc_pure_pursuit_plan(wp, w.shape[0], pp, lookahead, vgain, wheelbase, max_reacquire, out.ctypes.data_as(_dp))
    return out
