#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Build-container only (needs /root/reference).  The reference's Python is imported through
ref_loader (no-op numba shim), driven on seeded inputs, and its inputs/outputs are written
as small .npz fixtures.  No reference source is copied: the fixtures are data.

    python oracle/refshim/gen_golden.py            # regenerate everything
    python oracle/refshim/gen_golden.py scan ttc   # only some groups

Input map/raceline data files (example_map.{png,yaml}, example_waypoints.csv,
berlin/skirk.{png,yaml}, unittest/legacy_scan.npz) are copied verbatim as data fixtures.
"""
import os
import shutil
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

warnings.simplefilter("ignore")
REF = ref_loader.REF_ROOT
PARAM_KEYS = ['mu', 'C_Sf', 'C_Sr', 'lf', 'lr', 'h', 'm', 'I', 's_min', 's_max', 'sv_min',
              'sv_max', 'v_switch', 'a_max', 'v_min', 'v_max', 'width', 'length']
DEFAULT_PARAMS = {'mu': 1.0489, 'C_Sf': 4.718, 'C_Sr': 5.4562, 'lf': 0.15875, 'lr': 0.17145,
                  'h': 0.074, 'm': 3.74, 'I': 0.04712, 's_min': -0.4189, 's_max': 0.4189,
                  'sv_min': -3.2, 'sv_max': 3.2, 'v_switch': 7.319, 'a_max': 9.51,
                  'v_min': -5.0, 'v_max': 20.0, 'width': 0.31, 'length': 0.58}
EXAMPLE_MAP = os.path.join(GOLD, "maps", "example_map")


def pvec(p):
    return np.array([p[k] for k in PARAM_KEYS])


CHECK = {"on": False, "bad": [], "files": 0}


def save(name, **arrays):
    if CHECK["on"]:
        return check_against_committed(name, arrays)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("  wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.))


def check_against_committed(name, arrays):
    """--check: what the reference produces NOW (under whatever numba this interpreter has) against the committed fixture, key by
    key: integer / bool / string arrays must be equal; float arrays are reported exact or with their largest relative difference
    (<= 1e-9 passes: a compiled reference may differ from the interpreted one in libm ulps of sin / cos / tan / atan2)"""
    path = os.path.join(GOLD, name + ".npz")
    CHECK["files"] += 1
    if not os.path.isfile(path):
        CHECK["bad"].append("%s: no committed fixture" % name)
        print("  %-28s MISSING from tests/golden" % (name + ".npz"))
        return
    old = np.load(path)
    worst, exact, n = 0.0, 0, 0
    for key, val in arrays.items():
        val = np.asarray(val)
        n += 1
        if key not in old.files:
            CHECK["bad"].append("%s[%s]: not in the committed fixture" % (name, key))
            continue
        ref = old[key]
        if ref.shape != val.shape:
            CHECK["bad"].append("%s[%s]: shape %s, committed %s" % (name, key, val.shape, ref.shape))
            continue
        if val.dtype.kind == "f" and ref.dtype.kind == "f":
            if np.array_equal(val, ref, equal_nan=True):
                exact += 1
                continue
            with np.errstate(divide="ignore", invalid="ignore"):
                diff = np.abs(val - ref)
                rel = np.where(diff > 1e-12, diff / np.abs(ref), 0.0)
            m = float(np.nanmax(rel)) if rel.size else 0.0
            if not np.array_equal(np.isnan(val), np.isnan(ref)):
                m = float("inf")
            worst = max(worst, m)
            if not m <= 1e-9:
                CHECK["bad"].append("%s[%s]: floats differ by %.3g relative" % (name, key, m))
        else:
            if np.array_equal(val, ref):
                exact += 1
            else:
                CHECK["bad"].append("%s[%s]: %d of %d entries differ" % (name, key, int(np.sum(val != ref)), val.size))
    extra = sorted(set(old.files) - set(arrays))
    if extra:
        CHECK["bad"].append("%s: committed keys the generator no longer writes: %s" % (name, extra))
    print("  %-28s %3d keys: %3d exact, largest relative float difference %.3g" % (name + ".npz", n, exact, worst))


def copy_data():
    os.makedirs(os.path.join(GOLD, "maps"), exist_ok=True)
    pairs = [("examples/example_map.png", "maps/example_map.png"),
             ("examples/example_map.yaml", "maps/example_map.yaml"),
             ("examples/example_waypoints.csv", "maps/example_waypoints.csv"),
             ("gym/f110_gym/envs/maps/berlin.png", "maps/berlin.png"),
             ("gym/f110_gym/envs/maps/berlin.yaml", "maps/berlin.yaml"),
             ("gym/f110_gym/envs/maps/skirk.png", "maps/skirk.png"),
             ("gym/f110_gym/envs/maps/skirk.yaml", "maps/skirk.yaml"),
             ("gym/f110_gym/unittest/legacy_scan.npz", "legacy_scan.npz")]
    for src, dst in pairs:
        d = os.path.join(GOLD, dst)
        shutil.copyfile(os.path.join(REF, src), d)
        os.chmod(d, 0o644)
    print("  copied %d data files" % len(pairs))


def raceline():
    w = np.loadtxt(os.path.join(GOLD, "maps", "example_waypoints.csv"), delimiter=';', skiprows=3)
    return w  # columns s, x, y, psi, kappa, vx, ax


# ------------------------------------------------------------------------------ dynamics
def gen_dynamics(ns):
    dm = ns.dynamic_models
    rng = np.random.default_rng(101)
    n = 400
    P = DEFAULT_PARAMS
    p = pvec(P)
    x = np.empty((n, 7)); u = np.empty((n, 2))
    x[:, 0] = rng.uniform(-50, 50, n); x[:, 1] = rng.uniform(-50, 50, n)
    x[:, 2] = rng.uniform(-0.45, 0.45, n)
    x[:, 3] = np.where(rng.random(n) < 0.35, rng.uniform(-0.6, 0.6, n), rng.uniform(-5.5, 21, n))
    x[:, 4] = rng.uniform(-0.5, 7.0, n); x[:, 5] = rng.uniform(-3, 3, n)
    x[:, 6] = rng.uniform(-0.5, 0.5, n)
    u[:, 0] = rng.uniform(-4, 4, n); u[:, 1] = rng.uniform(-12, 12, n)
    # constraint edges
    x[0, 2] = P['s_min']; u[0, 0] = -1.0
    x[1, 2] = P['s_max']; u[1, 0] = 1.0
    x[2, 3] = P['v_min']; u[2, 1] = -1.0
    x[3, 3] = P['v_max']; u[3, 1] = 1.0
    x[4, 3] = 0.5; x[5, 3] = -0.5; x[6, 3] = 0.4999999; x[7, 3] = 0.0
    x[8, 3] = 9.0; u[8, 1] = 11.0   # above v_switch
    f_st = np.array([dm.vehicle_dynamics_st(x[i], u[i], *p[:16]) for i in range(n)])
    f_ks = np.array([dm.vehicle_dynamics_ks(x[i, :5], u[i], *p[:16]) for i in range(n)])
    # pid grid
    m = 300
    pin = np.empty((m, 4))
    pin[:, 0] = rng.uniform(-6, 21, m)      # desired speed
    pin[:, 1] = rng.uniform(-0.5, 0.5, m)   # desired steer
    pin[:, 2] = rng.uniform(-5, 20, m)      # current speed
    pin[:, 3] = rng.uniform(-0.42, 0.42, m)  # current steer
    pin[0, 1] = pin[0, 3] + 5e-5; pin[1, 1] = pin[1, 3] - 2e-4; pin[2, 2] = 0.0
    pin[3, 0] = pin[3, 2]
    pout = np.array([dm.pid(pin[i, 0], pin[i, 1], pin[i, 2], pin[i, 3], P['sv_max'], P['a_max'],
                            P['v_max'], P['v_min']) for i in range(m)])
    save("dynamics", params=p, x=x, u=u, f_st=f_st, f_ks=f_ks, pid_in=pin, pid_out=pout)


# --------------------------------------------------------------------------- update_pose
def _make_car(ns, integrator, lidar_dist=0.0, params=None):
    bc = ns.base_classes
    ref_loader.fresh_racecar_class(ns)
    car = bc.RaceCar(dict(params or DEFAULT_PARAMS), 12345, is_ego=True, time_step=0.01,
                     integrator=integrator, lidar_dist=lidar_dist)
    car.set_map(EXAMPLE_MAP + ".yaml", ".png")
    return car


def gen_update_pose(ns):
    """RaceCar.update_pose single steps from random states + a rollout (scan discarded:
    ScanSimulator2D.scan is replaced by a stub while stepping)."""
    bc = ns.base_classes
    rng = np.random.default_rng(202)
    out = {}
    for name, integ, ld in [("rk4", bc.Integrator.RK4, 0.0), ("euler", bc.Integrator.Euler, 0.0),
                            ("rk4_lidar", bc.Integrator.RK4, 0.275)]:
        car = _make_car(ns, integ, ld)
        poses_seen = []
        car.scan_simulator.scan = lambda pose, rng_, std_dev=0.01: (poses_seen.append(np.array(pose)) or np.zeros(1080))
        n = 120
        st0 = np.empty((n, 7))
        st0[:, 0] = rng.uniform(-20, 20, n); st0[:, 1] = rng.uniform(-20, 20, n)
        st0[:, 2] = rng.uniform(-0.4, 0.4, n)
        st0[:, 3] = np.where(rng.random(n) < 0.3, rng.uniform(-0.6, 0.6, n), rng.uniform(-4, 15, n))
        st0[:, 4] = rng.uniform(-0.1, 6.4, n); st0[:, 5] = rng.uniform(-2, 2, n)
        st0[:, 6] = rng.uniform(-0.3, 0.3, n)
        st0[0, 4] = 6.29; st0[0, 5] = 1.0; st0[1, 4] = 0.001; st0[1, 5] = -2.0  # yaw wrap cases
        buf0 = rng.uniform(-0.4, 0.4, (n, 2)); cnt0 = rng.integers(0, 3, n)
        act = np.stack([rng.uniform(-0.45, 0.45, n), rng.uniform(-3, 12, n)], axis=1)
        st1 = np.empty((n, 7)); buf1 = np.zeros((n, 2)); cnt1 = np.empty(n, dtype=np.int64)
        sp = np.empty((n, 3))
        for i in range(n):
            car.state = st0[i].copy()
            car.steer_buffer = buf0[i, :cnt0[i]].copy()   # index 0 = newest
            car.update_pose(act[i, 0], act[i, 1])
            st1[i] = car.state
            cnt1[i] = car.steer_buffer.shape[0]
            buf1[i, :cnt1[i]] = car.steer_buffer
            sp[i] = poses_seen[-1]
        out.update({name + "_state0": st0, name + "_buf0": buf0, name + "_cnt0": cnt0,
                    name + "_action": act, name + "_state1": st1, name + "_buf1": buf1,
                    name + "_cnt1": cnt1, name + "_scan_pose": sp})
        if name != "euler":
            # rollout from reset, 400 steps, piecewise-constant random actions
            car.reset(np.array([0.7, 0.0, 1.37079632679]))
            T = 400
            acts = np.empty((T, 2)); traj = np.empty((T, 7))
            a = np.zeros(2)
            for t in range(T):
                if t % 25 == 0:
                    a = np.array([rng.uniform(-0.3, 0.3), rng.uniform(0.5, 8.0)])
                acts[t] = a
                car.update_pose(a[0], a[1])
                traj[t] = car.state
            out.update({name + "_roll_actions": acts, name + "_roll_states": traj})
    ref_loader.fresh_racecar_class(ns)
    save("update_pose", params=pvec(DEFAULT_PARAMS), lidar_dist=np.array([0.0, 0.0, 0.275]), **out)


# ---------------------------------------------------------------------------------- scan
class ScanProbe(object):
    """Instrument laser_models so each trace_ray call reports the table index it used and
    the (r,c) of the sample that ended its loop (module globals are looked up at call time
    under the no-op njit shim)."""

    def __init__(self, lm):
        self.lm = lm
        self.orig_xy = lm.xy_2_rc
        self.orig_tr = lm.trace_ray
        self.last_rc = (0, 0); self.lookups = 0
        self.rcs = []; self.idx = []

    def __enter__(self):
        lm = self.lm

        def xy(*a):
            rc = self.orig_xy(*a)
            self.last_rc = rc; self.lookups += 1
            return rc

        def tr(x, y, theta_index, *rest):
            d = self.orig_tr(x, y, theta_index, *rest)
            self.rcs.append(self.last_rc); self.idx.append(int(theta_index))
            return d
        lm.xy_2_rc = xy; lm.trace_ray = tr
        return self

    def __exit__(self, *a):
        self.lm.xy_2_rc = self.orig_xy; self.lm.trace_ray = self.orig_tr


def _scan_cases(ns, map_yaml, num_beams, poses, fov=4.7):
    lm = ns.laser_models
    sim = lm.ScanSimulator2D(num_beams, fov)
    sim.set_map(map_yaml, ".png")
    scans = np.empty((len(poses), num_beams)); rcs = np.empty((len(poses), num_beams, 2), dtype=np.int32)
    idx = np.empty((len(poses), num_beams), dtype=np.int32); lookups = np.empty(len(poses), dtype=np.int64)
    for k, pose in enumerate(poses):
        with ScanProbe(lm) as pr:
            scans[k] = sim.scan(np.array(pose), None)
            rcs[k] = np.array(pr.rcs); idx[k] = np.array(pr.idx); lookups[k] = pr.lookups
    return sim, scans, rcs, idx, lookups


def gen_scan(ns):
    rng = np.random.default_rng(303)
    w = raceline()
    # example_map: raceline poses (heading psi+pi/2) + perturbed + far/off-map
    ks = (np.arange(14) * 57) % w.shape[0]
    poses = [[w[k, 1], w[k, 2], w[k, 3] + np.pi / 2] for k in ks]
    for k in ks[:6]:
        poses.append([w[k, 1] + rng.uniform(-0.4, 0.4), w[k, 2] + rng.uniform(-0.4, 0.4),
                      rng.uniform(0, 2 * np.pi)])
    poses += [[0.7, 0.0, 1.37079632679], [0.7, 0.0, 0.0], [0.7, 0.0, 7.5], [0.7, 0.0, -2.2],
              [-78.0, -44.0, 0.7],       # just inside the map corner, far from the track
              [-90.0, -50.0, 0.3],       # outside the map: OOB reads dt[-1,-1]
              [30.0, 70.0, 4.0]]         # outside on the other side
    poses = np.array(poses)
    sim, scans, rcs, idx, lk = _scan_cases(ns, EXAMPLE_MAP + ".yaml", 1080, poses)
    save("scan_example_map", poses=poses, scans=scans, hit_rc=rcs, dir_idx=idx, lookups=lk,
         dt_corner=np.array([sim.dt[-1, -1]]), dt_shape=np.array(sim.dt.shape),
         dt_checksum=np.array([sim.dt.sum(), (sim.dt * np.arange(sim.dt.shape[1])[None, :]).sum()]),
         sines=sim.sines, cosines=sim.cosines,
         theta_index_increment=np.array([sim.theta_index_increment]))
    # berlin: dt[-1,-1] == 0 -> rays end where they leave the map
    bposes = np.array([[0.0, 0.0, th] for th in np.linspace(-1, 1, 4)] +
                      [[0.5, -0.3, 2.5], [-1.0, 0.4, 5.0], [40.0, 40.0, 1.0]])
    bsim, bscans, brcs, bidx, blk = _scan_cases(ns, os.path.join(GOLD, "maps", "berlin.yaml"), 1080, bposes)
    save("scan_berlin", poses=bposes, scans=bscans, hit_rc=brcs, dir_idx=bidx, lookups=blk,
         dt_corner=np.array([bsim.dt[-1, -1]]), dt_shape=np.array(bsim.dt.shape),
         dt_checksum=np.array([bsim.dt.sum(), (bsim.dt * np.arange(bsim.dt.shape[1])[None, :]).sum()]))
    # 4096 beams (BASELINE config 5 shape; theta_dis stays 2000)
    p4 = poses[[0, 3, 14, 20]]
    s4, scans4, rcs4, idx4, lk4 = _scan_cases(ns, EXAMPLE_MAP + ".yaml", 4096, p4)
    save("scan_example_map_4096", poses=p4, scans=scans4, hit_rc=rcs4, dir_idx=idx4, lookups=lk4)
    # odd beam count / fov
    p3 = poses[[1, 15]]
    s3, scans3, rcs3, idx3, lk3 = _scan_cases(ns, EXAMPLE_MAP + ".yaml", 271, p3, fov=6.0)
    save("scan_example_map_271", poses=p3, scans=scans3, hit_rc=rcs3, dir_idx=idx3, lookups=lk3,
         fov=np.array([6.0]))


# ----------------------------------------------------------------------------------- ttc
def gen_ttc(ns):
    lm = ns.laser_models
    car = _make_car(ns, ns.base_classes.Integrator.RK4)
    rc = ns.base_classes.RaceCar
    sa, co, sd = rc.scan_angles.copy(), rc.cosines.copy(), rc.side_distances.copy()
    rng = np.random.default_rng(404)
    n = 60
    scans = np.empty((n, 1080)); vels = np.empty(n); flags = np.empty(n, dtype=np.int32)
    for i in range(n):
        base = rng.uniform(0.5, 10.0, 1080)
        kind = i % 6
        v = rng.uniform(0.5, 8.0)
        if kind == 0:
            pass
        elif kind == 1:      # one beam just inside the threshold ahead
            j = rng.integers(400, 680); base[j] = sd[j] + 0.5 * 0.005 * v * co[j]
        elif kind == 2:      # just outside
            j = rng.integers(400, 680); base[j] = sd[j] + 1.5 * 0.005 * v * co[j]
        elif kind == 3:      # reversing
            v = -rng.uniform(0.5, 4.0); j = rng.integers(0, 60); base[j] = sd[j] + 0.3 * 0.005 * v * co[j]
        elif kind == 4:
            v = 0.0; base[:] = 0.0
        else:                # scan below side distance: negative ttc -> no collision
            base = sd * 0.9
        scans[i] = base; vels[i] = v
        flags[i] = int(lm.check_ttc_jit(base, v, sa, co, sd, 0.005))
    ref_loader.fresh_racecar_class(ns)
    save("ttc", scan_angles=sa, cosines=co, side_distances=sd, scans=scans, vels=vels, flags=flags,
         width=np.array([DEFAULT_PARAMS['width']]), lf=np.array([DEFAULT_PARAMS['lf']]),
         lr=np.array([DEFAULT_PARAMS['lr']]))


# ----------------------------------------------------------------------------- collision
def gen_collision(ns):
    cm = ns.collision_models
    rng = np.random.default_rng(505)
    L, W = DEFAULT_PARAMS['length'], DEFAULT_PARAMS['width']
    n = 1500
    pa = np.stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(-4, 7, n)], axis=1)
    d = rng.uniform(0.0, 0.9, n); ang = rng.uniform(0, 2 * np.pi, n)
    pb = np.stack([pa[:, 0] + d * np.cos(ang), pa[:, 1] + d * np.sin(ang), rng.uniform(-4, 7, n)], axis=1)
    pb[:20] = pa[:20]                      # identical boxes (d == 0 seed direction case)
    pb[20:40, 2] = pa[20:40, 2]            # parallel boxes
    va = np.array([cm.get_vertices(pa[i], L, W) for i in range(n)])
    vb = np.array([cm.get_vertices(pb[i], L, W) for i in range(n)])
    flags = np.array([int(cm.collision(np.ascontiguousarray(va[i]), np.ascontiguousarray(vb[i]))) for i in range(n)], dtype=np.int32)
    # collision_multiple on groups of 5 bodies
    G = 60
    gp = np.stack([rng.uniform(-1, 1, (G, 5)), rng.uniform(-1, 1, (G, 5)), rng.uniform(0, 6.3, (G, 5))], axis=2)
    gcol = np.empty((G, 5)); gidx = np.empty((G, 5))
    for g in range(G):
        allv = np.array([cm.get_vertices(gp[g, a], L, W) for a in range(5)])
        gcol[g], gidx[g] = cm.collision_multiple(allv)
    save("collision", length=np.array([L]), width=np.array([W]), pose_a=pa, pose_b=pb,
         vert_a=va, vert_b=vb, flags=flags, group_poses=gp, group_collisions=gcol, group_idx=gidx)


# ------------------------------------------------------------------------------- raycast
def gen_raycast(ns):
    lm, cm = ns.laser_models, ns.collision_models
    car = _make_car(ns, ns.base_classes.Integrator.RK4)
    sa = ns.base_classes.RaceCar.scan_angles.copy()
    ref_loader.fresh_racecar_class(ns)
    rng = np.random.default_rng(606)
    L, W = DEFAULT_PARAMS['length'], DEFAULT_PARAMS['width']
    n = 90
    ego = np.stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(0, 2 * np.pi, n)], axis=1)
    dist = rng.uniform(0.35, 6.0, n); bearing = rng.uniform(-np.pi, np.pi, n)
    dist[:10] = rng.uniform(0.0, 0.25, 10)            # lidar inside / overlapping the box
    bearing[10:30] = np.pi + rng.uniform(-0.4, 0.4, 20)  # opponent behind (rear +-pi straddle)
    dist[10:30] = rng.uniform(0.4, 1.5, 20)
    ego[30:34, 2] = 0.0                                # ego heading zeroed by a wall hit
    opp = np.stack([ego[:, 0] + dist * np.cos(ego[:, 2] + bearing),
                    ego[:, 1] + dist * np.sin(ego[:, 2] + bearing), rng.uniform(0, 2 * np.pi, n)], axis=1)
    verts = np.array([cm.get_vertices(opp[i], L, W) for i in range(n)])
    base = 10.0
    lo = np.empty(n, dtype=np.int32); hi = np.empty(n, dtype=np.int32)
    scans = np.empty((n, 1080))
    for i in range(n):
        lo[i], hi[i] = lm.get_blocked_view_indices(ego[i], verts[i], sa)
        scans[i] = lm.ray_cast(ego[i], np.full(1080, base), sa, verts[i])
    # get_range unit cases incl. collinear / parallel
    gr_in = []
    for _ in range(200):
        p = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), 0.0]); bt = rng.uniform(-4, 8)
        va = rng.uniform(-3, 3, 2); vb = rng.uniform(-3, 3, 2)
        gr_in.append(np.concatenate([p, [bt], va, vb]))
    for k in range(12):   # beam exactly along the edge direction (collinear) and parallel offsets
        p = np.array([0.0, 0.0, 0.0]); bt = 0.0 if k % 2 == 0 else np.pi / 2
        dirv = np.array([np.cos(bt), np.sin(bt)]); off = 0.0 if k < 6 else 0.5
        va = dirv * (1.0 + k) + np.array([-dirv[1], dirv[0]]) * off
        vb = dirv * (3.0 + k) + np.array([-dirv[1], dirv[0]]) * off
        gr_in.append(np.concatenate([p, [bt], va, vb]))
    gr_in = np.array(gr_in)
    gr_out = np.array([lm.get_range(r[:3], r[3], r[4:6], r[6:8]) for r in gr_in])
    save("raycast", scan_angles=sa, length=np.array([L]), width=np.array([W]), ego=ego, opp=opp,
         vertices=verts, min_ind=lo, max_ind=hi, base=np.array([base]), scans=scans,
         get_range_in=gr_in, get_range_out=gr_out)


# --------------------------------------------------------------------------- sim rollout
def gen_sim(ns):
    """2-agent Simulator.step trajectory on example_map with the seed-12345 noise stream;
    the ego is driven into a wall (iTTC hit, heading zeroed) with the opponent in view."""
    bc = ns.base_classes
    ref_loader.fresh_racecar_class(ns)
    sim = bc.Simulator(dict(DEFAULT_PARAMS), 2, 12345, time_step=0.01, integrator=bc.Integrator.RK4)
    sim.set_map(EXAMPLE_MAP + ".yaml", ".png")
    w = raceline()
    k0 = 40
    start = np.array([[w[k0, 1], w[k0, 2], w[k0, 3] + np.pi / 2],
                      [w[k0 + 8, 1], w[k0 + 8, 2], w[k0 + 8, 3] + np.pi / 2]])  # opponent 1.6 m ahead
    sim.reset(start)
    T = 260
    rng = np.random.default_rng(707)
    acts = np.empty((T, 2, 2)); states = np.empty((T, 2, 7)); cols = np.empty((T, 2))
    incol = np.empty((T, 2), dtype=np.int32); cidx = np.empty((T, 2)); snap = np.empty((T, 2, 3))
    sub = np.empty((T, 2, 135)); ssum = np.empty((T, 2)); full_steps = [0, 1, 2, 60, 130, 200, 259]
    full = {}
    a = np.zeros((2, 2))
    for t in range(T):
        if t % 20 == 0:
            a = np.array([[rng.uniform(-0.15, 0.15), rng.uniform(4.0, 7.0)],
                          [rng.uniform(-0.05, 0.05), rng.uniform(0.5, 2.0)]])
        if t >= 120:
            a[0, 0] = 0.4   # steer the ego hard into the wall
        acts[t] = a
        obs = sim.step(a)
        states[t] = np.array([ag.state for ag in sim.agents])
        cols[t] = obs['collisions']; cidx[t] = sim.collision_idx
        incol[t] = [int(ag.in_collision) for ag in sim.agents]
        snap[t] = sim.agent_poses
        for i in range(2):
            sc = np.asarray(obs['scans'][i])
            sub[t, i] = sc[::8]; ssum[t, i] = sc.sum()
        if t in full_steps:
            full["scans_t%d" % t] = np.array(obs['scans'])
    noise = np.random.default_rng(12345).normal(0., 0.01, size=1080)
    ref_loader.fresh_racecar_class(ns)
    save("sim_rollout", params=pvec(DEFAULT_PARAMS), start=start, actions=acts, states=states,
         collisions=cols, in_collision=incol, collision_idx=cidx, agent_poses=snap,
         scans_sub8=sub, scans_sum=ssum, full_steps=np.array(full_steps),
         noise_row0=noise, seed=np.array([12345]), **full)
    print("    wall hits at steps:", np.nonzero(incol.any(axis=1))[0][:10], " gjk:", np.nonzero(cols.any(axis=1))[0][:5])


# ------------------------------------------------------------------- multi-agent sim rollout
def gen_sim_multi(ns, cases=((3, 40, 220, 110, 90), (4, 300, 220, 110, 90), (8, 520, 240, 110, 90)), name="sim_rollout_multi", variants=None):
    """Simulator.step with A = 3, 4 and 8 cars on example_map (base_classes.py:553-612 loops over A
    agents; collision_models.py:184-212 all pairs; base_classes.py:206-227 every opponent per ego).
    The cars start as a bunched train on the raceline (0.8 m apart, alternating lateral offsets, so that
    the opponent windows of an ego overlap and every car but the last has one exactly behind it, where
    the bearing wraps at +-pi); the rear cars are faster than the front ones (rear-end GJK contacts,
    several pairs at once) and one car in the middle is steered into the wall (iTTC hit, heading zeroed
    while the others still ray-cast it)."""
    bc = ns.base_classes
    w = raceline()
    out = {}
    for A, k0, T, wall_from, mid_full in cases:
        ref_loader.fresh_racecar_class(ns)
        var = (variants or {}).get(A, {})
        pdict = dict(DEFAULT_PARAMS); pdict.update(var.get("params", {}))
        sim = bc.Simulator(pdict, A, 12345, time_step=var.get("time_step", 0.01), integrator=getattr(bc.Integrator, var.get("integrator", "RK4")),
                           lidar_dist=var.get("lidar_dist", 0.0))
        sim.set_map(os.path.join(GOLD, "maps", var.get("map", "example_map")) + ".yaml", ".png")
        if var.get("agent_params"):   # Simulator.update_params(params, agent_idx) :503-519: a parameter set per agent slot
            rows = []
            for i in range(A):
                pi = dict(pdict); pi.update(var["agent_params"].get(i, {}))
                sim.update_params(pi, agent_idx=i)
                rows.append(pvec(pi))
            out["a%d_agent_params" % A] = np.array(rows)
        if variants is not None:
            out["a%d_map" % A] = np.array([var.get("map", "example_map")])
            out["a%d_time_step" % A] = np.array([var.get("time_step", 0.01)])
            out["a%d_params" % A] = pvec(pdict)
            out["a%d_integrator" % A] = np.array([{"RK4": 1, "Euler": 2}[var.get("integrator", "RK4")]])
            out["a%d_lidar_dist" % A] = np.array([var.get("lidar_dist", 0.0)])
        start = np.empty((A, 3))
        for i in range(A):
            k = (k0 + 4 * i) % w.shape[0]
            th = w[k, 3] + np.pi / 2
            lat = 0.12 * (1 if i % 2 else -1)
            start[i] = [w[k, 1] - lat * np.sin(th), w[k, 2] + lat * np.cos(th), th]
        if "start" in var:
            start = np.array(var["start"], dtype=float)
        sim.reset(start)
        rng = np.random.default_rng(900 + A)
        acts = np.empty((T, A, 2)); states = np.empty((T, A, 7)); cols = np.empty((T, A))
        incol = np.empty((T, A), dtype=np.int32); cidx = np.empty((T, A)); snap = np.empty((T, A, 3))
        sub = np.empty((T, A, 45)); ssum = np.empty((T, A))
        full_steps = [0, mid_full, T - 1]
        a = np.zeros((A, 2))
        waller = A // 2
        for t in range(T):
            if t % 25 == 0:
                for i in range(A):
                    a[i] = [rng.uniform(-0.06, 0.06), 6.5 - 5.0 * i / (A - 1) + rng.uniform(-0.4, 0.4)]
            if t >= wall_from:
                a[waller] = [0.41, 5.0]
            acts[t] = a
            obs = sim.step(a)
            states[t] = np.array([ag.state for ag in sim.agents])
            cols[t] = obs['collisions']; cidx[t] = sim.collision_idx
            incol[t] = [int(ag.in_collision) for ag in sim.agents]
            snap[t] = sim.agent_poses
            for i in range(A):
                sc = np.asarray(obs['scans'][i])
                sub[t, i] = sc[::24]; ssum[t, i] = sc.sum()
            if t in full_steps:
                out["a%d_scans_t%d" % (A, t)] = np.array(obs['scans'])
        out.update({"a%d_start" % A: start, "a%d_actions" % A: acts, "a%d_states" % A: states,
                    "a%d_collisions" % A: cols, "a%d_in_collision" % A: incol, "a%d_collision_idx" % A: cidx,
                    "a%d_agent_poses" % A: snap, "a%d_scans_sub24" % A: sub, "a%d_scans_sum" % A: ssum,
                    "a%d_full_steps" % A: np.array(full_steps)})
        gjk_steps = np.nonzero((cidx >= 0).any(axis=1))[0]
        print("    A=%d: wall hits at %s (agents %s); gjk contact steps %d (first %s), pairs seen %s" % (
            A, np.nonzero(incol.any(axis=1))[0][:4], np.nonzero(incol.any(axis=0))[0], len(gjk_steps), gjk_steps[:3],
            sorted({(int(i), int(j)) for row in cidx for i, j in enumerate(row) if j >= 0})[:10]))
    ref_loader.fresh_racecar_class(ns)
    save(name, params=pvec(DEFAULT_PARAMS), seed=np.array([12345]), agent_counts=np.array([c[0] for c in cases]), **out)


def gen_sim_variants(ns):
    """the same scenario away from the defaults every other Simulator fixture uses: 2 cars under the EULER integrator with the
    lidar 0.275 m ahead of the rear axle (base_classes.py:69 lidar_dist, :373-380), and 3 longer, wider, heavier cars on
    slipperier tyres (the opponent's box is drawn with the EGO's length / width, :223; GJK with the Simulator's, :549), and 4
    cars that all differ (Simulator.update_params per agent slot)."""
    gen_sim_multi(ns, cases=((2, 40, 200, 120, 80), (3, 300, 200, 110, 90), (4, 520, 200, 110, 90), (5, 0, 160, 60, 80), (6, 150, 120, 60, 50)), name="sim_rollout_variants",
                  variants={2: {"integrator": "Euler", "lidar_dist": 0.275},
                            3: {"lidar_dist": 0.275, "params": {"length": 0.72, "width": 0.40, "m": 4.2, "I": 0.06, "mu": 0.8, "lf": 0.18, "lr": 0.19}},
                            # four DIFFERENT cars (update_params per slot): every ego draws its opponents with its OWN length / width (:223),
                            # collision_multiple uses the Simulator's (:549)
                            4: {"agent_params": {0: {"length": 0.50, "width": 0.27, "m": 3.2}, 1: {"length": 0.70, "width": 0.38, "mu": 0.85},
                                                 3: {"length": 0.62, "width": 0.24, "a_max": 7.5}}},
                            # another track: berlin (resolution 0.05 — not a power of two; the table's last cell is 0, so rays end where
                            # they leave the map), five cars fanning out from a loose cluster in its free middle
                            5: {"map": "berlin", "start": [[0.0, 0.0, 0.3], [0.9, 0.5, 2.0], [-0.8, 0.6, 4.1], [0.4, -0.9, 5.3], [-0.7, -0.8, 1.1]]},
                            # a coarser clock: time_step 0.02 (integration, steer-delay FIFO and iTTC all see it), six cars
                            6: {"time_step": 0.02}})


def gen_sim_many(ns):
    """the same scenario with 12 and 20 cars (round 4: the product steps 9 .. 16 cars per env with the 256-record form of
    k_finalize_multi and more with k_finalize_multi_tiled; until now those two were pinned to the oracle only).  Shorter
    runs — the un-jitted reference needs seconds per step at this size — with the wall hit moved forward accordingly."""
    gen_sim_multi(ns, cases=((12, 100, 130, 45, 60), (20, 610, 120, 40, 60)), name="sim_rollout_many")


# ------------------------------------------------------------------- env, 2 agents, ego_idx = 1
def gen_env2(ns, three=False, kw=None):
    """(three=True: gen_env3 below.)  F110Env(num_agents=2, ego_idx=1) — the reference's default agent count (f110_env.py:133-136)
    with the ego in slot 1 — through three episodes on example_map, each started by env.reset:
      0  both cars leave the start zone and reverse back into it twice at different speeds, with
         different start headings (the zone of BOTH cars is laid out in the EGO's start frame,
         f110_env.py:331): toggles of both cars, done only when both reach 4 (:244);
      1  car 0 (not the ego) is steered into the wall first — collisions[0] = 1 and the episode goes
         on — then the ego hits the wall: done on collisions[ego_idx] (:244);
      2  car 0 rear-ends the ego: GJK sets both flags, done."""
    ns = ref_loader.load_reference(with_env=True)
    ref_loader.fresh_racecar_class(ns)
    if kw is not None:   # gen_env_kwargs: every constructor keyword away from its default
        env = ns.f110_env.F110Env(map=EXAMPLE_MAP, map_ext='.png', **kw)
    else:
        env = ns.f110_env.F110Env(map=EXAMPLE_MAP, map_ext='.png', num_agents=3 if three else 2, ego_idx=2 if three else 1, seed=12345)
    w = raceline()
    obs_ego = []
    keys = ("x", "y", "th", "v", "w", "lap_time", "lap_count", "done", "toggle", "near", "col", "ckpt", "scan_sum")
    out = {}

    def pose_at(k, lat=0.0, dth=0.0):
        th = w[k, 3] + np.pi / 2
        return [w[k, 1] - lat * np.sin(th), w[k, 2] + lat * np.cos(th), th + dth]

    def run(ep, start, policy, max_steps):
        rec = {k: [] for k in keys}
        acts = []

        def log(obs, done, info):
            rec["x"].append(list(obs['poses_x'])); rec["y"].append(list(obs['poses_y']))
            rec["th"].append(list(obs['poses_theta'])); rec["v"].append(list(obs['linear_vels_x']))
            rec["w"].append(list(obs['ang_vels_z']))
            rec["lap_time"].append(np.array(obs['lap_times'], dtype=float).copy())
            rec["lap_count"].append(np.array(obs['lap_counts'], dtype=float).copy())
            rec["done"].append(bool(done)); rec["toggle"].append(np.array(env.toggle_list, dtype=float).copy())
            rec["near"].append(np.array(env.near_starts, dtype=bool).copy())
            rec["col"].append(np.array(obs['collisions'], dtype=float).copy())
            rec["ckpt"].append(np.array(info['checkpoint_done'], dtype=bool).copy())
            rec["scan_sum"].append([float(np.sum(s)) for s in obs['scans']])
        start = np.array(start)
        obs, r, done, info = env.reset(start.copy())
        obs_ego.append(obs['ego_idx'])   # F110Env does not hand ego_idx to its Simulator (f110_env.py:192): the obs says 0
        log(obs, done, info)
        t = 0
        while t < max_steps and not done:
            a = policy(t, env)
            acts.append(a.copy())
            obs, r, done, info = env.step(a)
            log(obs, done, info)
            t += 1
        out["ep%d_start" % ep] = start
        out["ep%d_actions" % ep] = np.array(acts)
        for k in keys:
            out["ep%d_%s" % (ep, k)] = np.array(rec[k])
        print("    episode %d: %d steps, toggles %s, collisions %s, done %s" % (ep, t, env.toggle_list, rec["col"][-1], done))
        return rec

    if three:
        # episode 0: laps of three cars — two side by side, the ego 2 m behind them; done only when ALL three have 4 toggles
        def laps3(t, env):
            a = np.zeros((3, 2))
            for i, sp in ((0, 1.8), (1, 2.6), (2, 2.2)):
                a[i, 1] = sp if env.toggle_list[i] % 2 == 0 else -sp
            return a
        rec = run(0, [pose_at(0, lat=0.45, dth=-0.5), pose_at(0, lat=1.45), pose_at(w.shape[0] - 10, lat=0.95)], laps3, 4000)
        print("      collision steps:", np.nonzero(np.array(rec["col"]).any(axis=1))[0][:10])
        assert rec["done"][-1] and not np.any(np.array(rec["col"])) and np.all(rec["toggle"][-1] >= 4)
        fin = [int(np.argmax(np.array(rec["toggle"])[:, i] >= 4)) for i in range(3)]
        assert len(set(fin)) == 3, fin

        # episode 1: the two non-ego cars hit the wall one after the other, the episode goes on; then the ego does
        def walls3(t, env):
            return np.array([[0.41 if t >= 40 else 0.0, 5.0], [0.41 if t >= 120 else 0.0, 4.5], [0.2 if t >= 220 else 0.0, 4.0]])
        rec = run(1, [pose_at(120), pose_at(160), pose_at(200)], walls3, 1500)
        col = np.array(rec["col"])
        first = [int(np.argmax(col[:, i] > 0)) for i in range(3)]
        assert col[:, 0].any() and col[:, 1].any() and first[0] < first[1] < first[2] and rec["done"][-1] and not rec["done"][first[1]], first

        # episode 2: car 0 rear-ends car 1 (two non-egos: flags set, the episode goes on), then reaches the slow ego ahead
        def ram3(t, env):
            return np.array([[0.0, 7.0], [0.0, 3.0], [0.0, 1.0]])
        rec = run(2, [pose_at(400), pose_at(408), pose_at(440)], ram3, 1500)
        col = np.array(rec["col"])
        first01 = int(np.argmax((col[:, 0] > 0) & (col[:, 1] > 0)))
        assert rec["done"][-1] and col[-1, 2] == 1 and 0 < first01 < len(col) - 1 and not rec["done"][first01], first01
        ref_loader.fresh_racecar_class(ns)
        assert set(obs_ego) == {0}
        save("env_episode_3agents", ego_idx=np.array([2]), obs_ego_idx=np.array([obs_ego[0]]), seed=np.array([12345]), **out)
        return

    # episode 0: laps.  the cars sit side by side (1 m apart, mid-track); car 0's heading is 0.5 rad off the ego's
    def laps(t, env):
        a = np.zeros((2, 2))
        for i, sp in ((0, 1.8), (1, 2.6)):
            a[i, 1] = sp if env.toggle_list[i] % 2 == 0 else -sp
        return a
    rec = run(0, [pose_at(0, lat=0.45, dth=-0.5), pose_at(0, lat=1.45)], laps, 4000)
    print("      collision steps:", np.nonzero(np.array(rec["col"]).any(axis=1))[0][:10])
    assert rec["done"][-1] and not np.any(np.array(rec["col"])) and np.all(rec["toggle"][-1] >= 4)
    first_both = [int(np.argmax(np.array(rec["toggle"])[:, i] >= 4)) for i in range(2)]
    assert first_both[0] != first_both[1], "cars should finish at different steps"

    # episode 1: the non-ego car hits the wall first, the ego later
    def walls(t, env):
        return np.array([[0.41 if t >= 40 else 0.0, 5.0], [0.41 if t >= 200 else 0.0, 4.0]])
    rec = run(1, [pose_at(120, lat=0.0), pose_at(160, lat=0.0)], walls, 1500)
    col = np.array(rec["col"])
    first0 = int(np.argmax(col[:, 0] > 0)); first1 = int(np.argmax(col[:, 1] > 0))
    assert col[:, 0].any() and first0 < first1 and rec["done"][-1] and not rec["done"][first0], (first0, first1)

    if kw is not None:
        ref_loader.fresh_racecar_class(ns)
        save("env_episode_kwargs", ego_idx=np.array([kw["ego_idx"]]), obs_ego_idx=np.array([obs_ego[0]]), seed=np.array([kw["seed"]]),
             timestep=np.array([kw["timestep"]]), lidar_dist=np.array([kw["lidar_dist"]]), integrator=np.array([kw["integrator"].value]),
             params=pvec(kw["params"]), **out)
        return

    # episode 2: car 0 rear-ends the ego
    def ram(t, env):
        return np.array([[0.0, 7.0], [0.0, 1.0]])
    rec = run(2, [pose_at(400, lat=0.0), pose_at(408, lat=0.0)], ram, 1500)
    col = np.array(rec["col"])
    assert rec["done"][-1] and col[-1, 0] == 1 and col[-1, 1] == 1
    ref_loader.fresh_racecar_class(ns)
    assert set(obs_ego) == {0}
    save("env_episode_2agents", ego_idx=np.array([1]), obs_ego_idx=np.array([obs_ego[0]]), seed=np.array([12345]), **out)


def gen_env_kwargs(ns):
    """F110Env with every constructor keyword away from its default (f110_env.py:103-160): seed 999, timestep 0.02, the Euler
    integrator, the lidar 0.2 m ahead of the rear axle, other vehicle parameters — two episodes (laps of both cars; a non-ego
    wall hit, then the ego's).  Pins the keyword plumbing of the drop-in F110Env / F110VecEnv, lap times in units of the
    new timestep included."""
    ns2 = ref_loader.load_reference(with_env=True)
    p = dict(DEFAULT_PARAMS); p.update({"mu": 0.9, "m": 3.9, "length": 0.60, "width": 0.33, "a_max": 8.0, "v_max": 15.0})
    gen_env2(ns, kw={"num_agents": 2, "ego_idx": 1, "seed": 999, "timestep": 0.02, "integrator": ns2.base_classes.Integrator.Euler,
                     "lidar_dist": 0.2, "params": p})


def gen_env_updates(ns):
    """F110Env.update_params (one agent, then all: f110_env.py:364-375) in the middle of an episode and F110Env.update_map
    (:351-362) between two: 2 agents on example_map — 40 steps, agent 1 gets other parameters, 40 steps, every agent gets a
    third set, then the ego is steered into the wall — then the SAME env object moves to berlin for a second episode."""
    ns = ref_loader.load_reference(with_env=True)
    ref_loader.fresh_racecar_class(ns)
    env = ns.f110_env.F110Env(map=EXAMPLE_MAP, map_ext='.png', num_agents=2, seed=12345)
    w = raceline()
    p1 = dict(DEFAULT_PARAMS); p1.update({"mu": 0.7, "m": 4.4, "a_max": 6.0, "length": 0.66, "width": 0.36})
    p2 = dict(DEFAULT_PARAMS); p2.update({"mu": 1.2, "C_Sf": 5.4, "C_Sr": 6.1, "a_max": 9.0, "v_max": 12.0, "sv_max": 2.2})
    keys = ("x", "y", "th", "v", "w", "lap_time", "lap_count", "done", "toggle", "near", "col", "ckpt", "scan_sum")
    out = {"upd_steps": np.array([40, 80]), "upd_index": np.array([1, -1]), "upd_params": np.array([pvec(p1), pvec(p2)]),
           "ep1_map": np.array(["berlin"])}

    def run(ep, start, policy, max_steps, hooks):
        rec = {k: [] for k in keys}
        acts = []

        def log(obs, done, info):
            rec["x"].append(list(obs['poses_x'])); rec["y"].append(list(obs['poses_y'])); rec["th"].append(list(obs['poses_theta']))
            rec["v"].append(list(obs['linear_vels_x'])); rec["w"].append(list(obs['ang_vels_z']))
            rec["lap_time"].append(np.array(obs['lap_times'], dtype=float).copy()); rec["lap_count"].append(np.array(obs['lap_counts'], dtype=float).copy())
            rec["done"].append(bool(done)); rec["toggle"].append(np.array(env.toggle_list, dtype=float).copy())
            rec["near"].append(np.array(env.near_starts, dtype=bool).copy()); rec["col"].append(np.array(obs['collisions'], dtype=float).copy())
            rec["ckpt"].append(np.array(info['checkpoint_done'], dtype=bool).copy()); rec["scan_sum"].append([float(np.sum(sc)) for sc in obs['scans']])
        start = np.array(start, dtype=float)
        obs, r, done, info = env.reset(start.copy())
        log(obs, done, info)
        t = 0
        while t < max_steps and not done:
            if t in hooks:
                hooks[t]()
            a = policy(t)
            acts.append(a.copy())
            obs, r, done, info = env.step(a)
            log(obs, done, info)
            t += 1
        out["ep%d_start" % ep] = start
        out["ep%d_actions" % ep] = np.array(acts)
        for k in keys:
            out["ep%d_%s" % (ep, k)] = np.array(rec[k])
        print("    episode %d: %d steps, collisions %s, done %s" % (ep, t, rec["col"][-1], done))
        return rec

    def th_at(k):
        return w[k, 3] + np.pi / 2
    rec = run(0, [[w[120, 1], w[120, 2], th_at(120)], [w[160, 1], w[160, 2], th_at(160)]],
              lambda t: np.array([[0.2 if t >= 120 else 0.03 * np.sin(t / 9.0), 5.0], [0.02 * np.cos(t / 7.0), 4.0]]), 1200,
              {40: lambda: env.update_params(p1, index=1), 80: lambda: env.update_params(p2)})
    assert rec["done"][-1] and rec["col"][-1][0] == 1 and len(rec["done"]) > 100
    env.update_map(os.path.join(GOLD, "maps", "berlin.yaml"), ".png")
    rec = run(1, [[0.0, 0.0, 0.3], [0.9, 0.5, 2.0]], lambda t: np.array([[0.05, 6.0], [-0.05, 3.0]]), 1200, {})
    assert rec["done"][-1] and rec["col"][-1][0] == 1
    ref_loader.fresh_racecar_class(ns)
    save("env_episode_updates", seed=np.array([12345]), **out)


def gen_env3(ns):
    """F110Env(num_agents=3, ego_idx=2): the done rule over MORE than two agents (f110_env.py:244: every agent's toggles,
    the ego's collision only) — laps of three cars finishing at three different steps, two non-ego wall hits that end
    nothing before the ego's does, a contact between the two non-egos that ends nothing before one of them reaches the ego."""
    gen_env2(ns, three=True)


# ----------------------------------------------------------------------------------- env
def gen_env(ns):
    """F110Env episode (1 agent, example_map): drive out of the start zone and reverse back
    into it twice -> toggles 0..4, lap_counts, lap_times, done (f110_env.py:204-246)."""
    ns = ref_loader.load_reference(with_env=True)
    ref_loader.fresh_racecar_class(ns)
    env = ns.f110_env.F110Env(map=EXAMPLE_MAP, map_ext='.png', num_agents=1, seed=12345)
    start = np.array([[0.7, 0.0, 1.37079632679]])
    obs, r, done, info = env.reset(start)
    rec = {k: [] for k in ("x", "y", "th", "v", "lap_time", "lap_count", "done", "toggle", "near", "col", "scan_sum")}
    acts = []

    def log(obs, done):
        rec["x"].append(obs['poses_x'][0]); rec["y"].append(obs['poses_y'][0])
        rec["th"].append(obs['poses_theta'][0]); rec["v"].append(obs['linear_vels_x'][0])
        rec["lap_time"].append(float(obs['lap_times'][0])); rec["lap_count"].append(float(obs['lap_counts'][0]))
        rec["done"].append(bool(done)); rec["toggle"].append(float(env.toggle_list[0]))
        rec["near"].append(bool(env.near_starts[0])); rec["col"].append(float(obs['collisions'][0]))
        rec["scan_sum"].append(float(np.sum(obs['scans'][0])))
    log(obs, done)
    t = 0
    phase_speed = 2.5
    while t < 3000 and not done:
        tog = env.toggle_list[0]
        # forward until out of the zone (odd toggle), then reverse until back in (even)
        speed = phase_speed if tog % 2 == 0 else -phase_speed
        a = np.array([[0.0, speed]])
        acts.append(a[0].copy())
        obs, r, done, info = env.step(a)
        log(obs, done)
        t += 1
    print("    env episode: %d steps, toggles=%s done=%s" % (t, env.toggle_list, done))
    ref_loader.fresh_racecar_class(ns)
    save("env_episode", start=start, actions=np.array(acts), **{k: np.array(v) for k, v in rec.items()})


# ------------------------------------------------------------------------- waypoint_follow
def gen_waypoint_follow(ns):
    """BASELINE configs[0]: examples/waypoint_follow.py — the reference's PurePursuitPlanner drives
    one car around example_map in the reference F110Env until done (two laps).  Records the
    planner's actions (so the replay needs no planner) and the resulting trajectory."""
    import importlib.util
    import types
    from argparse import Namespace
    import yaml
    ns = ref_loader.load_reference(with_env=True)
    sys.modules["pyglet.gl"].GL_POINTS = 0
    spec = importlib.util.spec_from_file_location("waypoint_follow_ref", os.path.join(REF, "examples", "waypoint_follow.py"))
    wf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wf)
    with open(os.path.join(REF, "examples", "config_example_map.yaml")) as f:
        conf = Namespace(**yaml.safe_load(f))
    conf.wpt_path = os.path.join(GOLD, "maps", "example_waypoints.csv")
    planner = wf.PurePursuitPlanner(conf, (0.17145 + 0.15875))
    work = {'tlad': 0.82461887897713965, 'vgain': 1.375}
    ref_loader.fresh_racecar_class(ns)
    env = ns.f110_env.F110Env(map=EXAMPLE_MAP, map_ext='.png', num_agents=1, timestep=0.01,
                              integrator=ns.base_classes.Integrator.RK4)
    start = np.array([[conf.sx, conf.sy, conf.stheta]])
    obs, r, done, info = env.reset(start)
    acts, rec = [], []
    t = 0
    while not done and t < 7000:
        speed, steer = planner.plan(obs['poses_x'][0], obs['poses_y'][0], obs['poses_theta'][0], work['tlad'], work['vgain'])
        a = np.array([[steer, speed]])
        obs, r, done, info = env.step(a)
        acts.append(a[0].copy())
        rec.append([obs['poses_x'][0], obs['poses_y'][0], obs['poses_theta'][0], obs['linear_vels_x'][0],
                    obs['ang_vels_z'][0], float(obs['lap_times'][0]), float(obs['lap_counts'][0]),
                    float(obs['collisions'][0]), float(done), float(np.sum(obs['scans'][0]))])
        t += 1
    rec = np.array(rec)
    print("    waypoint_follow: %d steps, laps %.0f, lap time %.2f, collided %s, max v %.2f"
          % (t, rec[-1, 6], rec[-1, 5], bool(rec[:, 7].any()), rec[:, 3].max()))
    ref_loader.fresh_racecar_class(ns)
    save("waypoint_follow", start=start, actions=np.array(acts), traj=rec,
         columns=np.array(["x", "y", "theta", "v", "yaw_rate", "lap_time", "lap_count", "collision", "done", "scan_sum"]))


# ------------------------------------------------------------------------- planner
def gen_planner(ns):
    """examples/waypoint_follow.py:15-217 — PurePursuitPlanner.plan and its helpers as functions of
    the pose: nearest point (index, distance, t), look-ahead waypoint index and the (speed, steer)
    the planner returns.  Poses: the golden lap's trajectory (every 7th step), the same poses pushed
    off the raceline (re-acquire branch, 0.82 m < distance < 20 m), far away (> 20 m: default
    action) and exactly on waypoints."""
    import importlib.util
    from argparse import Namespace
    import yaml
    ref_loader.load_reference(with_env=True)
    sys.modules["pyglet.gl"].GL_POINTS = 0
    spec = importlib.util.spec_from_file_location("waypoint_follow_ref2", os.path.join(REF, "examples", "waypoint_follow.py"))
    wf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wf)
    with open(os.path.join(REF, "examples", "config_example_map.yaml")) as f:
        conf = Namespace(**yaml.safe_load(f))
    conf.wpt_path = os.path.join(GOLD, "maps", "example_waypoints.csv")
    wheelbase = 0.17145 + 0.15875
    planner = wf.PurePursuitPlanner(conf, wheelbase)
    tlad, vgain = 0.82461887897713965, 1.375
    lap = np.load(os.path.join(GOLD, "waypoint_follow.npz"))["traj"][::7, :3]
    rng = np.random.default_rng(2024)
    off = lap[::3].copy()
    off[:, 0] += rng.uniform(-4, 4, off.shape[0]); off[:, 1] += rng.uniform(-4, 4, off.shape[0]); off[:, 2] += rng.uniform(-1, 1, off.shape[0])
    far = lap[::40].copy() + np.array([60.0, -45.0, 0.3])
    wp = planner.waypoints
    onwp = np.stack([wp[::25, conf.wpt_xind], wp[::25, conf.wpt_yind], wp[::25, 3] + np.pi / 2], axis=1)
    near = lap[::5].copy()
    near[:, 0] += rng.uniform(-0.5, 0.5, near.shape[0]); near[:, 1] += rng.uniform(-0.5, 0.5, near.shape[0]); near[:, 2] += rng.uniform(-0.6, 0.6, near.shape[0])
    poses = np.concatenate([lap, off, far, onwp, near], axis=0)
    wpts = np.vstack((wp[:, conf.wpt_xind], wp[:, conf.wpt_yind])).T
    out, near_rec, look_rec = [], [], []
    for x, y, th in poses:
        speed, steer = planner.plan(x, y, th, tlad, vgain)
        out.append([steer, speed])
        pos = np.array([x, y])
        npnt, ndist, t, i = wf.nearest_point_on_trajectory(pos, wpts)
        near_rec.append([float(i), float(ndist), float(t)])
        i2 = -2          # -2: not searched / nothing found; -1 is a real answer (the closing segment)
        if ndist < tlad:
            _, i2_, _ = wf.first_point_on_trajectory_intersecting_circle(pos, tlad, wpts, i + t, wrap=True)
            i2 = -2 if i2_ is None else int(i2_)
        look_rec.append(i2)
    out = np.array(out)
    print("    planner: %d poses; default-action %d, re-acquire %d, look-ahead %d" % (
        len(poses), int(np.sum((out[:, 1] == 4.0) & (out[:, 0] == 0.0))),
        int(np.sum((np.array(near_rec)[:, 1] >= tlad) & (np.array(near_rec)[:, 1] < 20.0))), int(np.sum(np.array(look_rec) >= -1))))
    save("planner", poses=poses, actions=out, nearest=np.array(near_rec), lookahead_index=np.array(look_rec, dtype=np.int64),
         waypoints=np.stack([wp[:, conf.wpt_xind], wp[:, conf.wpt_yind], wp[:, conf.wpt_vind]], axis=1),
         tlad=np.array([tlad]), vgain=np.array([vgain]), wheelbase=np.array([wheelbase]), max_reacquire=np.array([20.0]))



# ------------------------------------------------- round 5: the rest of the public surface
PKG_MAPS = os.path.join(REPO, "f1tenth_gym_amd", "maps")   # the package's copy of the reference's tracks (vegas, stata_basement ...)


def _map_yaml(name):
    for d in (os.path.join(GOLD, "maps"), PKG_MAPS):
        if os.path.isfile(os.path.join(d, name + ".yaml")) and os.path.isfile(os.path.join(d, name + ".png")):
            return os.path.join(d, name + ".yaml")
    raise FileNotFoundError(name)


def _variant_yaml(tmp, name, resolution, origin):
    """the track's image under ANOTHER resolution / origin (yaw included): yaml + png pair in a scratch directory
    (the tests rebuild the same pair from the numbers stored in the fixture)"""
    src = os.path.splitext(_map_yaml(name))[0] + ".png"
    stem = os.path.join(tmp, "%s_variant" % name)
    shutil.copyfile(src, stem + ".png")
    with open(stem + ".yaml", "w") as f:
        f.write("image: %s.png\nresolution: %r\norigin: [%r, %r, %r]\nnegate: 0\noccupied_thresh: 0.65\nfree_thresh: 0.196\n"
                % (os.path.basename(stem), resolution, origin[0], origin[1], origin[2]))
    return stem + ".yaml"


def _scan_cases_ctor(ns, map_yaml, poses, num_beams, fov, **ctor):
    lm = ns.laser_models
    sim = lm.ScanSimulator2D(num_beams, fov, **ctor)
    sim.set_map(map_yaml, ".png")
    scans = np.empty((len(poses), num_beams)); rcs = np.empty((len(poses), num_beams, 2), dtype=np.int32)
    idx = np.empty((len(poses), num_beams), dtype=np.int32); lookups = np.empty(len(poses), dtype=np.int64)
    for k, pose in enumerate(poses):
        with ScanProbe(lm) as pr:
            scans[k] = sim.scan(np.array(pose), None)
            rcs[k] = np.array(pr.rcs); idx[k] = np.array(pr.idx); lookups[k] = pr.lookups
    return sim, scans, rcs, idx, lookups


def _poses_in_free_space(sim, rng, n, clearance=0.3):
    """world poses over free cells close to walls (on the track or just outside it) of a loaded reference ScanSimulator2D
    (any origin yaw); the last one lies outside the image"""
    rr, cc = np.nonzero((sim.dt > clearance) & (sim.dt < 1.2))
    pick = rng.integers(0, len(rr), n)
    u = (cc[pick] + rng.uniform(0.0, 1.0, n)) * sim.map_resolution
    v = (rr[pick] + rng.uniform(0.0, 1.0, n)) * sim.map_resolution
    c, s = sim.orig_c, sim.orig_s
    poses = np.stack([sim.orig_x + c * u - s * v, sim.orig_y + s * u + c * v, rng.uniform(-7.0, 7.0, n)], axis=1)
    poses[0::2, 2] = rng.uniform(0.0, 2 * np.pi, len(poses[0::2]))   # headings update_pose's yaw wrap leaves alone (the step-path tests use these)
    poses[-1, :2] = [sim.orig_x - 3.0 * c + 2.0 * s, sim.orig_y - 3.0 * s - 2.0 * c]
    return poses


SCAN_CTOR_CASES = [
    # map, num_beams, fov, eps, theta_dis, max_range  (laser_models.py:360-381: every keyword away from its default)
    ("example_map", 1080, 4.7, 0.03, 720, 8.0),      # 720 directions < 1080 beams: theta_index_increment 0.4992 < 1
    ("example_map", 541, 4.7, 0.2, 3600, 30.0),      # eps above the resolution (0.0625): the march ends on NON-zero table values
    ("berlin", 1080, 4.7, 0.2, 2000, 8.0),           # eps = 4 cells of berlin (0.05), short max_range; the table's last cell is 0
    ("berlin", 360, 6.2, 0.0001, 720, 30.0),         # fov close to a full turn, coarse direction table
    ("skirk", 777, 3.3, 0.03, 3600, 12.5),
    ("example_map", 2048, 4.7, 0.0001, 1000, 30.0),  # twice as many beams as directions
    ("skirk", 1500, 6.28, 0.0001, 1000, 30.0),       # fov ~ a full turn on 1000 directions: the last beams land on beam 0's directions again
]


def gen_scan_ctor(ns):
    """ScanSimulator2D(num_beams, fov, eps, theta_dis, max_range) away from the defaults + scan(pose, rng, std_dev != 0.01)
    (laser_models.py:360-381, :429-454)."""
    import tempfile
    out = {}
    for k, (name, beams, fov, eps, theta_dis, max_range) in enumerate(SCAN_CTOR_CASES):
        rng = np.random.default_rng(1100 + k)
        lm = ns.laser_models
        probe = lm.ScanSimulator2D(beams, fov, eps=eps, theta_dis=theta_dis, max_range=max_range)
        probe.set_map(_map_yaml(name), ".png")
        poses = _poses_in_free_space(probe, rng, 5)
        if fov > 6.2:    # the full-turn case: every heading one the step-path tests can use (no yaw wrap in update_pose)
            poses[:, 2] = rng.uniform(0.0, 2 * np.pi, len(poses))
        sim, scans, rcs, idx, lk = _scan_cases_ctor(ns, _map_yaml(name), poses, beams, fov, eps=eps, theta_dis=theta_dis, max_range=max_range)
        std = [0.05, 0.2, 0.001][k % 3]
        noisy = sim.scan(np.array(poses[0]), np.random.default_rng(4242 + k), std_dev=std)
        out.update({"c%d_map" % k: np.array([name]), "c%d_ctor" % k: np.array([beams, fov, eps, theta_dis, max_range]),
                    "c%d_poses" % k: poses, "c%d_scans" % k: scans, "c%d_hit_rc" % k: rcs, "c%d_dir_idx" % k: idx, "c%d_lookups" % k: lk,
                    "c%d_theta_index_increment" % k: np.array([sim.theta_index_increment]),
                    "c%d_noise_seed_std" % k: np.array([4242 + k, std]), "c%d_noisy" % k: noisy})
        nz = int(np.sum(sim.dt[np.clip(rcs[..., 0], 0, sim.dt.shape[0] - 1), np.clip(rcs[..., 1], 0, sim.dt.shape[1] - 1)] > 0))
        print("    case %d %-12s beams %4d eps %g theta_dis %d max_range %g: lookups %s, rays ending on a non-zero cell %d, at max_range %d"
              % (k, name, beams, eps, theta_dis, max_range, lk, nz, int(np.sum(scans >= max_range))))
    save("scan_ctor_variants", n_cases=np.array([len(SCAN_CTOR_CASES)]), **out)


ROTATED_CASES = [
    # base track, resolution, origin [x, y, yaw]   (laser_models.py:55-86 xy_2_rc rotates by -yaw; :417-420 orig_s / orig_c)
    ("berlin", 0.07, [-3.0, -4.0, 0.3]),
    ("skirk", 0.0437, [1.5, 2.5, -1.1]),
    ("example_map", 0.11, [-40.0, -20.0, 2.4]),
]


def gen_scan_rotated(ns):
    """yaml files whose origin has a yaw and whose resolution is not a power of two: scans from poses over free space, and a
    2-car Simulator rollout (noise seed 12345, a wall hit) on the first of them."""
    import tempfile
    out = {}
    bc = ns.base_classes
    with tempfile.TemporaryDirectory() as tmp:
        for k, (name, res2, org2) in enumerate(ROTATED_CASES):
            rng = np.random.default_rng(1200 + k)
            y = _variant_yaml(tmp, name, res2, org2)
            probe = ns.laser_models.ScanSimulator2D(1080, 4.7)
            probe.set_map(y, ".png")
            poses = _poses_in_free_space(probe, rng, 6)
            sim, scans, rcs, idx, lk = _scan_cases_ctor(ns, y, poses, 1080, 4.7)
            out.update({"r%d_map" % k: np.array([name]), "r%d_resolution" % k: np.array([res2]), "r%d_origin" % k: np.array(org2),
                        "r%d_poses" % k: poses, "r%d_scans" % k: scans, "r%d_hit_rc" % k: rcs, "r%d_dir_idx" % k: idx, "r%d_lookups" % k: lk})
            print("    rotated %d %-12s res %g yaw %g: lookups %s" % (k, name, res2, org2[2], lk))
        # Simulator rollout on the rotated berlin: two cars side by side in the free middle, car 0 is steered into the wall
        name, res2, org2 = ROTATED_CASES[0]
        y = _variant_yaml(tmp, name, res2, org2)
        ref_loader.fresh_racecar_class(ns)
        sim = bc.Simulator(dict(DEFAULT_PARAMS), 2, 12345, time_step=0.01, integrator=bc.Integrator.RK4)
        sim.set_map(y, ".png")
        ss = bc.RaceCar.scan_simulator
        rr, cc = np.nonzero((ss.dt > 1.2) & (ss.dt < 1.4))
        u0, v0 = (cc[0] + 0.5) * res2, (rr[0] + 0.5) * res2
        c, s = ss.orig_c, ss.orig_s
        to_world = lambda u, v: [ss.orig_x + c * u - s * v, ss.orig_y + s * u + c * v]
        start = np.array([to_world(u0, v0) + [org2[2] + 0.2], to_world(u0 + 0.3, v0 + 0.9) + [org2[2] + 0.5]])
        sim.reset(start.copy())
        T = 220
        rng = np.random.default_rng(1299)
        acts = np.empty((T, 2, 2)); states = np.empty((T, 2, 7)); cols = np.empty((T, 2)); incol = np.empty((T, 2), dtype=np.int32)
        cidx = np.empty((T, 2)); sub = np.empty((T, 2, 45)); ssum = np.empty((T, 2)); full = {}
        a = np.zeros((2, 2))
        for t in range(T):
            if t % 25 == 0:
                a = np.array([[rng.uniform(-0.1, 0.1), rng.uniform(3.0, 6.0)], [rng.uniform(-0.2, 0.2), rng.uniform(1.0, 3.0)]])
            if t >= 60:
                a[0] = [0.04, 6.0]      # (full lock would circle inside the free space: radius 0.76 m)
            acts[t] = a
            obs = sim.step(a.copy())
            states[t] = np.array([ag.state for ag in sim.agents]); cols[t] = obs['collisions']; cidx[t] = sim.collision_idx
            incol[t] = [int(ag.in_collision) for ag in sim.agents]
            for i in range(2):
                sc = np.asarray(obs['scans'][i]); sub[t, i] = sc[::24]; ssum[t, i] = sc.sum()
            if t in (0, 70, T - 1):
                full["sim_scans_t%d" % t] = np.array(obs['scans'])
        print("    rollout on rotated %s: wall hits at %s, contacts %d" % (name, np.nonzero(incol.any(axis=1))[0][:5], int((cidx >= 0).any(axis=1).sum())))
        assert incol.any()
        ref_loader.fresh_racecar_class(ns)
        out.update({"sim_start": start, "sim_actions": acts, "sim_states": states, "sim_collisions": cols, "sim_in_collision": incol,
                    "sim_collision_idx": cidx, "sim_scans_sub24": sub, "sim_scans_sum": ssum, "sim_full_steps": np.array([0, 70, T - 1]),
                    "sim_seed": np.array([12345]), **full})
    save("scan_rotated_origin", n_cases=np.array([len(ROTATED_CASES)]), params=pvec(DEFAULT_PARAMS), **out)


def gen_env_defaults(ns):
    """`F110Env()` with NO keyword at all (f110_env.py:104-159): the vegas track that ships inside the package, 2 agents, ego_idx 0,
    seed 12345, RK4, timestep 0.01.  Episode 0: both cars leave the start zone and reverse back into it twice (done on
    toggles); episode 1: car 1 (not the ego) hits the wall, the episode goes on until the ego does."""
    ns = ref_loader.load_reference(with_env=True)
    ref_loader.fresh_racecar_class(ns)
    env = ns.f110_env.F110Env()
    assert env.map_path.endswith("maps/vegas.yaml") and env.num_agents == 2 and env.ego_idx == 0 and env.seed == 12345
    ss = ns.base_classes.RaceCar.scan_simulator
    # a start on the track: the widest free spot of the lower-left straight; heading = the longest beam of a scan from there
    dt = ss.dt
    sub = dt[300:700, 100:600]
    r0, c0 = np.unravel_index(np.argmax(sub), sub.shape)
    x0, y0 = ss.orig_x + (100 + c0 + 0.5) * ss.map_resolution, ss.orig_y + (300 + r0 + 0.5) * ss.map_resolution
    look = ns.laser_models.ScanSimulator2D(720, 2 * np.pi * 719 / 720)
    look.orig_x, look.orig_y, look.orig_c, look.orig_s = ss.orig_x, ss.orig_y, ss.orig_c, ss.orig_s
    look.map_height, look.map_width, look.map_resolution, look.dt = ss.map_height, ss.map_width, ss.map_resolution, ss.dt
    sc = look.scan(np.array([x0, y0, 0.0]), None)
    th = -np.pi * 719 / 720 + np.argmax(sc) * look.angle_increment
    left = np.array([-np.sin(th), np.cos(th)])
    keys = ("x", "y", "th", "v", "w", "lap_time", "lap_count", "done", "toggle", "near", "col", "ckpt", "scan_sum")
    out = {}
    obs_ego = []

    def run(ep, start, policy, max_steps):
        rec = {k: [] for k in keys}
        acts = []

        def log(obs, done, info):
            rec["x"].append(list(obs['poses_x'])); rec["y"].append(list(obs['poses_y'])); rec["th"].append(list(obs['poses_theta']))
            rec["v"].append(list(obs['linear_vels_x'])); rec["w"].append(list(obs['ang_vels_z']))
            rec["lap_time"].append(np.array(obs['lap_times'], dtype=float).copy()); rec["lap_count"].append(np.array(obs['lap_counts'], dtype=float).copy())
            rec["done"].append(bool(done)); rec["toggle"].append(np.array(env.toggle_list, dtype=float).copy())
            rec["near"].append(np.array(env.near_starts, dtype=bool).copy()); rec["col"].append(np.array(obs['collisions'], dtype=float).copy())
            rec["ckpt"].append(np.array(info['checkpoint_done'], dtype=bool).copy()); rec["scan_sum"].append([float(np.sum(s_)) for s_ in obs['scans']])
        start = np.array(start, dtype=float)
        obs, r, done, info = env.reset(start.copy())
        obs_ego.append(obs['ego_idx'])
        log(obs, done, info)
        t = 0
        while t < max_steps and not done:
            a = policy(t)
            acts.append(a.copy())
            obs, r, done, info = env.step(a)
            log(obs, done, info)
            t += 1
        out["ep%d_start" % ep] = start
        out["ep%d_actions" % ep] = np.array(acts)
        for k in keys:
            out["ep%d_%s" % (ep, k)] = np.array(rec[k])
        print("    episode %d: %d steps, toggles %s, collisions %s, done %s" % (ep, t, env.toggle_list, rec["col"][-1], done))
        return rec

    def laps(t):
        a = np.zeros((2, 2))
        for i, sp in ((0, 2.4), (1, 1.7)):
            a[i, 1] = sp if env.toggle_list[i] % 2 == 0 else -sp
        return a
    start = [[x0 - 0.4 * left[0], y0 - 0.4 * left[1], th], [x0 + 0.5 * left[0], y0 + 0.5 * left[1], th + 0.15]]
    rec = run(0, start, laps, 3000)
    assert rec["done"][-1] and not np.any(np.array(rec["col"])) and np.all(rec["toggle"][-1] >= 4)
    rec = run(1, start, lambda t: np.array([[0.08 if t >= 150 else 0.0, 4.0], [0.41 if t >= 30 else 0.0, 5.0]]), 1500)
    col = np.array(rec["col"])
    first = [int(np.argmax(col[:, i] > 0)) for i in range(2)]
    print("      first wall hits:", first)
    assert col[:, 1].any() and first[1] < first[0] and rec["done"][-1] and not rec["done"][first[1]]
    ref_loader.fresh_racecar_class(ns)
    assert set(obs_ego) == {0}
    save("env_episode_defaults", map=np.array(["vegas"]), ego_idx=np.array([0]), obs_ego_idx=np.array([0]), seed=np.array([12345]), **out)


# ------------------------------------------------------------------- the rest of `from f110_gym.envs import *`
def gen_star_exports(ns):
    """Every function the reference's envs/__init__.py:2-5 star-exports that had no fixture of its own: the two constraint
    functions, the GJK helpers, get_trmtx, cross / are_collinear, xy_2_rc / distance_transform / trace_ray / get_scan as FREE
    functions (their own argument lists), get_dt, get_blocked_view_indices — and ray_cast / get_blocked_view_indices /
    check_ttc_jit with scan_angles tables that are NOT the uniform ramp (two concatenated ramps, a jittered ramp, a descending
    ramp, a short random table): the reference's argmin over the whole table decides there."""
    lm, cm, dm = ns.laser_models, ns.collision_models, ns.dynamic_models
    rng = np.random.default_rng(909)
    out = {}
    # ---- dynamic_models.py:29-87 (rows straddle every branch: limits reached, v above v_switch, exact ties)
    n = 400
    a_in = np.stack([rng.uniform(-7, 25, n), rng.uniform(-15, 15, n), rng.uniform(5, 9, n), rng.uniform(5, 12, n), np.full(n, -5.0), np.full(n, 20.0)], axis=1)
    a_in[:20, 0] = -5.0; a_in[20:40, 0] = 20.0; a_in[40:50, 1] = -a_in[40:50, 3]; a_in[50:60, 1] = a_in[50:60, 3]; a_in[60:70, 0] = a_in[60:70, 2]
    out["accl_in"], out["accl_out"] = a_in, np.array([dm.accl_constraints(*r) for r in a_in])
    s_in = np.stack([rng.uniform(-0.6, 0.6, n), rng.uniform(-5, 5, n), np.full(n, -0.4189), np.full(n, 0.4189), np.full(n, -3.2), np.full(n, 3.2)], axis=1)
    s_in[:20, 0] = -0.4189; s_in[20:40, 0] = 0.4189; s_in[40:50, 1] = -3.2; s_in[50:60, 1] = 3.2; s_in[60:70, 1] = 0.0
    out["steer_in"], out["steer_out"] = s_in, np.array([dm.steering_constraint(*r) for r in s_in])
    # ---- laser_models.py:219-247
    c_in = rng.uniform(-3, 3, (n, 4))
    out["cross_in"], out["cross_out"] = c_in, np.array([lm.cross(r[:2], r[2:]) for r in c_in])
    col_in = rng.uniform(-3, 3, (n, 6))
    t = rng.uniform(-2, 3, n)
    col_in[:150, 4:6] = col_in[:150, 0:2] + t[:150, None] * (col_in[:150, 2:4] - col_in[:150, 0:2])            # exactly on the line (up to rounding)
    col_in[150:200, 4:6] += 0.0
    col_in[100:150, 5] += rng.choice([1e-9, 3e-9, 1e-8, 3e-8, 1e-7], 50)                                      # around the 1e-8 tolerance
    out["collinear_in"] = col_in
    out["collinear_out"] = np.array([float(lm.are_collinear(r[0:2].copy(), r[2:4].copy(), r[4:6].copy())) for r in col_in])
    # ---- collision_models.py:34-110, 218-235
    p_in = rng.uniform(-3, 3, (n, 2))
    out["perp_in"], out["perp_out"] = p_in, np.array([cm.perpendicular(r.copy()) for r in p_in])
    tp_in = rng.uniform(-3, 3, (n, 6))
    out["triple_in"], out["triple_out"] = tp_in, np.array([cm.tripleProduct(r[0:2].copy(), r[2:4].copy(), r[4:6].copy()) for r in tp_in])
    L, W = DEFAULT_PARAMS['length'], DEFAULT_PARAMS['width']
    poses = np.stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(-4, 7, n)], axis=1)
    va = np.array([cm.get_vertices(q, L, W) for q in poses])
    vb = np.array([cm.get_vertices(q + np.array([0.3, -0.2, 0.7]), L, W) for q in poses])
    d = rng.uniform(-1, 1, (n, 2))
    d[:40] = 0.0; d[40:80, 0] = 0.0; d[80:120] = (va[80:120, 1] - va[80:120, 0])          # ties: zero direction, axis-aligned, along an edge's normal partner
    out["body_a"], out["body_b"], out["dir"] = va, vb, d
    out["avg_out"] = np.array([cm.avgPoint(np.ascontiguousarray(v)) for v in va])
    out["furthest_out"] = np.array([cm.indexOfFurthestPoint(np.ascontiguousarray(va[i]), np.ascontiguousarray(d[i])) for i in range(n)], dtype=np.int32)
    out["support_out"] = np.array([cm.support(np.ascontiguousarray(va[i]), np.ascontiguousarray(vb[i]), np.ascontiguousarray(d[i])) for i in range(n)])
    pent = rng.uniform(-2, 2, (60, 5, 2))                                                    # bodies that are not quadrilaterals
    out["pent"], out["pent_dir"] = pent, d[:60] + 0.1
    out["pent_avg"] = np.array([cm.avgPoint(np.ascontiguousarray(v)) for v in pent])
    out["pent_furthest"] = np.array([cm.indexOfFurthestPoint(np.ascontiguousarray(pent[i]), np.ascontiguousarray(out["pent_dir"][i])) for i in range(60)], dtype=np.int32)
    out["trmtx_in"], out["trmtx_out"] = poses, np.array([cm.get_trmtx(q) for q in poses])
    # ---- laser_models.py:40-186 as free functions, on a small synthetic map with a yawed origin
    H, Wd, res = 96, 128, 0.05
    bitmap = np.full((H, Wd), 255.0)
    bitmap[0, :] = bitmap[-1, :] = bitmap[:, 0] = bitmap[:, -1] = 0.0
    bitmap[30:40, 50:70] = 0.0; bitmap[60:62, 10:100] = 0.0
    for _ in range(25):
        r, c = rng.integers(2, H - 2), rng.integers(2, Wd - 2)
        bitmap[r, c] = 0.0
    dt = lm.get_dt(bitmap, res)
    out["bitmap"], out["map_res"], out["dt"] = bitmap.astype(np.uint8), np.array([res]), dt
    for tag, (ox, oy, oth) in (("a", (-1.3, 0.7, 0.0)), ("b", (0.4, -2.1, 0.6))):
        oc, os_ = np.cos(oth), np.sin(oth)
        m = 300
        # world points: inside the map (cell coordinates -> world through the origin's rotation), plus some outside
        u = np.stack([rng.uniform(-0.4, Wd * res + 0.4, m), rng.uniform(-0.4, H * res + 0.4, m)], axis=1)
        u[:40, 0] = np.round(u[:40, 0] / res) * res                                   # on cell boundaries
        x = ox + oc * u[:, 0] - os_ * u[:, 1]; y = oy + os_ * u[:, 0] + oc * u[:, 1]
        out["pts_" + tag] = np.stack([x, y], axis=1)
        out["origin_" + tag] = np.array([ox, oy, oth])
        out["rc_" + tag] = np.array([lm.xy_2_rc(x[i], y[i], ox, oy, oc, os_, H, Wd, res) for i in range(m)], dtype=np.int32)
        out["dtval_" + tag] = np.array([lm.distance_transform(x[i], y[i], ox, oy, oc, os_, H, Wd, res, dt) for i in range(m)])
        theta_dis = 720
        th = np.linspace(0.0, 2 * np.pi, num=theta_dis)
        sines, cosines = np.sin(th), np.cos(th)
        tidx = rng.uniform(0, theta_dis - 1e-6, m)
        out["theta_idx_" + tag] = tidx
        out["trace_" + tag] = np.array([lm.trace_ray(x[i], y[i], tidx[i], sines, cosines, 1e-4, ox, oy, oc, os_, H, Wd, res, dt, 4.0) for i in range(m)])
        nb, fov = 181, 4.2
        inc = theta_dis * (fov / (nb - 1)) / (2. * np.pi)
        sp = np.stack([x[40:60], y[40:60], rng.uniform(0, 2 * np.pi, 20)], axis=1)
        out["scan_poses_" + tag] = sp
        out["scans_" + tag] = np.array([lm.get_scan(q, theta_dis, fov, nb, inc, sines, cosines, 1e-4, ox, oy, oc, os_, H, Wd, res, dt, 4.0) for q in sp])
    out["scan_cfg"] = np.array([720, 181, 4.2, 1e-4, 4.0])
    # ---- scan_angles tables that are not the uniform ramp (laser_models.py:282-346, 188-217)
    tables = {"two_ramps": np.concatenate([np.linspace(-2.35, 0.5, 90), np.linspace(-0.5, 2.35, 110)]),
              "jitter": np.linspace(-2.35, 2.35, 200) + rng.uniform(-0.03, 0.03, 200),
              "descending": np.linspace(2.35, -2.35, 200),
              "short_random": rng.uniform(-3.1, 3.1, 37),
              "uniform_ref": np.array([-4.7 / 2. + i * (4.7 / 199) for i in range(200)])}
    for name, sa in tables.items():
        B = sa.shape[0]
        m = 40
        ego = np.stack([rng.uniform(-5, 5, m), rng.uniform(-5, 5, m), rng.uniform(0, 2 * np.pi, m)], axis=1)
        dist = rng.uniform(0.35, 5.0, m); bearing = rng.uniform(-np.pi, np.pi, m)
        dist[:6] = rng.uniform(0.0, 0.25, 6); bearing[6:20] = np.pi + rng.uniform(-0.4, 0.4, 14)
        opp = np.stack([ego[:, 0] + dist * np.cos(ego[:, 2] + bearing), ego[:, 1] + dist * np.sin(ego[:, 2] + bearing), rng.uniform(0, 2 * np.pi, m)], axis=1)
        verts = np.array([cm.get_vertices(q, L, W) for q in opp])
        lo = np.empty(m, dtype=np.int32); hi = np.empty(m, dtype=np.int32); scans = np.empty((m, B))
        for i in range(m):
            lo[i], hi[i] = lm.get_blocked_view_indices(ego[i], verts[i], sa)
            scans[i] = lm.ray_cast(ego[i], np.full(B, 10.0), sa, verts[i])
        co = np.cos(sa); sd = np.abs(0.3 / np.maximum(np.abs(np.sin(sa)), 0.2))
        mt = 24
        tsc = rng.uniform(0.3, 8.0, (mt, B)); vel = rng.uniform(-3, 8, mt); vel[:3] = 0.0
        for i in range(0, mt, 3):
            j = rng.integers(0, B); tsc[i, j] = sd[j] + 0.4 * 0.005 * vel[i] * co[j]
        flags = np.array([int(lm.check_ttc_jit(tsc[i], vel[i], sa, co, sd, 0.005)) for i in range(mt)], dtype=np.int32)
        out.update({"sa_" + name: sa, "ego_" + name: ego, "verts_" + name: verts, "lo_" + name: lo, "hi_" + name: hi, "rc_scans_" + name: scans,
                    "ttc_scans_" + name: tsc, "ttc_vel_" + name: vel, "ttc_cos_" + name: co, "ttc_side_" + name: sd, "ttc_flags_" + name: flags})
    out["tables"] = np.array(sorted(tables))
    save("star_exports", **out)


GROUPS = {"star_exports": gen_star_exports, "scan_ctor": gen_scan_ctor, "scan_rotated": gen_scan_rotated, "env_defaults": gen_env_defaults, "sim_variants": gen_sim_variants, "sim_many": gen_sim_many, "planner": gen_planner, "data": lambda ns: copy_data(), "dynamics": gen_dynamics, "update_pose": gen_update_pose,
          "scan": gen_scan, "ttc": gen_ttc, "collision": gen_collision, "raycast": gen_raycast,
          "sim": gen_sim, "sim_multi": gen_sim_multi, "env": gen_env, "env2": gen_env2, "env3": gen_env3, "env_kwargs": gen_env_kwargs, "env_updates": gen_env_updates, "waypoint_follow": gen_waypoint_follow}


def main(argv):
    os.makedirs(GOLD, exist_ok=True)
    if "--check" in argv:
        # regenerate in memory and DIFF against tests/golden instead of writing: for a maintainer whose interpreter has a working
        # numba (this image has none) — the reference's @njit functions then run compiled, and the diff shows whether the
        # interpreted runs that produced the committed fixtures differ from the jitted path (tests/golden/README.md)
        argv = [a for a in argv if a != "--check"]
        CHECK["on"] = True
        os.environ["F110_REAL_NUMBA"] = "1"
    which = [g for g in (argv or list(GROUPS)) if not (CHECK["on"] and g == "data")]
    # the reference's example scripts import f110_gym.envs.* by name: bind those names to the reference
    # for the duration of the generation only (ref_loader.reference_modules)
    with ref_loader.reference_modules() as ns:
        for g in which:
            t = time.time()
            print("[%s]" % g)
            GROUPS[g](ns)
            print("  %.1f s" % (time.time() - t))
    if CHECK["on"]:
        print("--check: reference decorated with %s; %d fixture file(s) compared, %d problem(s)" % (ref_loader.numba_in_use(), CHECK["files"], len(CHECK["bad"])))
        for b in CHECK["bad"]:
            print("  " + b)
        return 1 if CHECK["bad"] else 0
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
