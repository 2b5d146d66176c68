"""Randomised parity fuzzing of the unit entry points (C ABI) against the CPU oracle."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _util import load_map_image, oracle_map_dt
from oracle import orc
import f1tenth_gym_amd as amd

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
P = amd.DEFAULT_PARAMS
pv = orc.params_vec(P)


def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    with np.errstate(invalid="ignore", divide="ignore"):   # |a - b| <= tol*|b| + 1e-12 (tests/_util.rel_err)
        ex = np.maximum(np.abs(a - b) - 1e-12, 0.0)
        d = np.where(ex > 0.0, ex / np.abs(b), 0.0)
    same_nan = np.isnan(a) & np.isnan(b)
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    d = np.where(same_nan | same_inf, 0.0, d)
    d = np.where(np.isnan(d), np.inf, d)
    return d.max() if d.size else 0.0


u = amd.BatchSim(num_envs=1, num_agents=1)
# ---- dynamics / pid / update_pose
n = 20000
x = np.stack([rng.uniform(-100, 100, n), rng.uniform(-100, 100, n), rng.uniform(-0.5, 0.5, n),
              np.where(rng.random(n) < 0.4, rng.uniform(-0.7, 0.7, n), rng.uniform(-6, 25, n)),
              rng.uniform(-20, 20, n), rng.uniform(-30, 30, n), rng.uniform(-1.5, 1.5, n)], axis=1)
x[:50, 3] = 0.0; x[50:100, 3] = 0.5; x[100:150, 3] = -0.5; x[150:200, 2] = P['s_max']; x[200:250, 2] = P['s_min']
uu = np.stack([rng.uniform(-5, 5, n), rng.uniform(-15, 15, n)], axis=1)
f_st, f_ks = u.dynamics_batch(x, uu, P)
r_st = np.array([orc.vehicle_dynamics_st(a, b, pv) for a, b in zip(x[:4000], uu[:4000])])
r_ks = np.array([orc.vehicle_dynamics_ks(a[:5], b, pv) for a, b in zip(x[:4000], uu[:4000])])
print("dynamics_st rel", rel(f_st[:4000], r_st), "ks", rel(f_ks[:4000], r_ks))
pin = np.stack([rng.uniform(-6, 22, n), rng.uniform(-0.5, 0.5, n), rng.uniform(-6, 22, n), rng.uniform(-0.45, 0.45, n)], axis=1)
pin[:100, 1] = pin[:100, 3] + rng.uniform(-2e-4, 2e-4, 100); pin[100:200, 2] = 0.0; pin[200:300, 0] = pin[200:300, 2]
po = u.pid_batch(pin, P)
pr = np.array([orc.pid(r[0], r[1], r[2], r[3], P['sv_max'], P['a_max'], P['v_max'], P['v_min']) for r in pin[:4000]])
print("pid exact", np.array_equal(po[:4000], pr))
for integ in (1, 2):
    cnt0 = rng.integers(0, 3, n).astype(np.int32); buf0 = rng.uniform(-0.4, 0.4, (n, 2)); act = np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-6, 22, n)], axis=1)
    s1, b1, c1, sp = u.update_pose_batch(x, buf0, cnt0, act, P, 0.01, integ, 0.3)
    worst = 0.0
    for i in range(3000):
        st, sb, cnt, spo = orc.update_pose(x[i], buf0[i], cnt0[i], act[i, 0], act[i, 1], pv, 0.01, integ, 0.3)
        worst = max(worst, rel(s1[i], st), rel(sp[i], spo)); assert cnt == c1[i] and np.array_equal(sb[:cnt], b1[i, :cnt])
    print("update_pose integ", integ, "rel", worst)
# ---- vertices / gjk / collision_multiple
m = 30000
pa = np.stack([rng.uniform(-3, 3, m), rng.uniform(-3, 3, m), rng.uniform(-10, 10, m)], axis=1)
d = np.where(rng.random(m) < 0.2, 0.0, rng.uniform(0, 1.0, m)); ang = rng.uniform(0, 6.28, m)
pb = np.stack([pa[:, 0] + d * np.cos(ang), pa[:, 1] + d * np.sin(ang), np.where(rng.random(m) < 0.2, pa[:, 2], rng.uniform(-10, 10, m))], axis=1)
va = u.get_vertices_batch(pa, 0.58, 0.31); vb = u.get_vertices_batch(pb, 0.58, 0.31)
rv = np.array([orc.get_vertices(p, 0.58, 0.31) for p in pa[:3000]])
print("vertices rel", rel(va[:3000], rv))
ova = np.array([orc.get_vertices(p, 0.58, 0.31) for p in pa]); ovb = np.array([orc.get_vertices(p, 0.58, 0.31) for p in pb])
fl = u.gjk_batch(ova, ovb)
rf = np.array([int(orc.collision(a, b)) for a, b in zip(ova, ovb)])
print("gjk mismatches", int(np.sum(fl != rf)), "of", m, "colliding", int(rf.sum()))
quads_a = rng.normal(size=(5000, 4, 2)); quads_b = rng.normal(size=(5000, 4, 2)) + rng.uniform(-1, 1, (5000, 1, 2))
fl = u.gjk_batch(quads_a, quads_b); rf = np.array([int(orc.collision(a, b)) for a, b in zip(quads_a, quads_b)])
print("gjk arbitrary quads mismatches", int(np.sum(fl != rf)), "colliding", int(rf.sum()))
G = 400; gp = np.stack([rng.uniform(-1, 1, (G, 6)), rng.uniform(-1, 1, (G, 6)), rng.uniform(0, 6.3, (G, 6))], axis=2)
allv = np.array([[orc.get_vertices(p, 0.58, 0.31) for p in g] for g in gp])
col, idx = u.collision_multiple_batch(allv)
ok = all(np.array_equal(col[g], orc.collision_multiple(allv[g])[0]) and np.array_equal(idx[g], orc.collision_multiple(allv[g])[1]) for g in range(G))
print("collision_multiple exact", ok)
# ---- ttc / raycast / get_range
B = 1080
k = 3000
scans = rng.uniform(0.0, 3.0, (k, B)); vels = np.where(rng.random(k) < 0.1, 0.0, rng.uniform(-5, 20, k))
j = rng.integers(0, B, k); scans[np.arange(k), j] = u.side_distances[j] + 0.005 * vels * u.cosines[j] * rng.uniform(0.0, 2.0, k)
fl = u.ttc_batch(scans, vels, 0.005)
rf = np.array([int(orc.check_ttc(s, v, u.cosines, u.side_distances, 0.005)) for s, v in zip(scans, vels)])
print("ttc mismatches", int(np.sum(fl != rf)), "hits", int(rf.sum()))
k = 4000
ego = np.stack([rng.uniform(-5, 5, k), rng.uniform(-5, 5, k), np.where(rng.random(k) < 0.1, 0.0, rng.uniform(-7, 7, k))], axis=1)
ego[:40, 2] = rng.choice([1e15, -4e12, 3e7, 1e5], 40)
dist = np.where(rng.random(k) < 0.15, rng.uniform(0, 0.4, k), rng.uniform(0.3, 15, k)); bear = np.where(rng.random(k) < 0.3, np.pi + rng.uniform(-0.6, 0.6, k), rng.uniform(-np.pi, np.pi, k))
opp = np.stack([ego[:, 0] + dist * np.cos(ego[:, 2] + bear), ego[:, 1] + dist * np.sin(ego[:, 2] + bear), rng.uniform(0, 6.28, k)], axis=1)
verts = np.array([orc.get_vertices(p, 0.58, 0.31) for p in opp])
base = rng.uniform(0.2, 20.0, (k, B))
out, mm = u.raycast_batch(ego, verts, base)
bad = 0; worst = 0.0
for i in range(k):
    r = orc.ray_cast(ego[i], base[i], u.scan_angles, verts[i])
    lo, hi = orc.get_blocked_view_indices(ego[i], verts[i], u.scan_angles)
    touched_same = np.array_equal(out[i] != base[i], r != base[i])
    e = rel(out[i], r)
    if (lo, hi) != tuple(mm[i]) or not touched_same or e > 1e-9:
        bad += 1
        if bad <= 3:
            print("  raycast mismatch case", i, "ego", ego[i], "win", (lo, hi), tuple(mm[i]), "touched_same", touched_same, "rel", e)
    worst = max(worst, e)
print("raycast bad cases", bad, "of", k, "worst rel", worst)
gi = np.concatenate([rng.uniform(-3, 3, (5000, 3)), rng.uniform(-8, 8, (5000, 1)), rng.uniform(-3, 3, (5000, 4))], axis=1)
go = u.get_range_batch(gi); gr = np.array([orc.get_range(r[:3], r[3], r[4:6], r[6:8]) for r in gi])
print("get_range inf-pattern equal", np.array_equal(np.isinf(go), np.isinf(gr)), "rel", rel(go, gr))
# ---- scans on random poses over the maps
for mapname in ("example_map", "berlin", "skirk"):
    img, res, origin = load_map_image(mapname); dt, _, _ = oracle_map_dt(mapname)
    H, W = dt.shape
    for layout in (0, 3):
        s = amd.BatchSim(num_envs=1, num_agents=1, map_layout=layout); s.set_map_image(img, res, origin)
        so = orc.ScanOracle(1080, 4.7); so.set_map_dt(dt, res, origin)
        kk = 300
        poses = np.stack([origin[0] + rng.uniform(-0.1, 1.1, kk) * W * res, origin[1] + rng.uniform(-0.1, 1.1, kk) * H * res, rng.uniform(-20, 20, kk)], axis=1)
        ranges, hits, lk = s.scan_batch(poses, want_hits=True, want_lookups=True)
        nb = 0
        for i in range(kk):
            r, h = so.scan(poses[i], want_hits=True)
            if not (np.array_equal(ranges[i], r) and np.array_equal(hits[i], h) and lk[i] == so.last_lookups):
                nb += 1
        print("scan", mapname, "layout", layout, "bad poses", nb, "of", kk)
        s.close()
# ---- round 5: the same with the ScanSimulator2D constructor arguments and the origin's yaw drawn too (laser_models.py:360-381, :417-420)
for trial in range(10):
    mapname = str(rng.choice(["example_map", "berlin", "skirk"]))
    img, res, origin = load_map_image(mapname); dt, _, _ = oracle_map_dt(mapname)
    H, W = dt.shape
    B = int(rng.choice([1080, 271, 64, 1500, 2048])); fov = float(rng.choice([4.7, 6.28, 3.0]))
    eps = float(rng.choice([1e-4, 0.03, 0.2])); theta_dis = int(rng.choice([2000, 720, 1000, 3600])); max_range = float(rng.choice([30.0, 8.0, 12.5]))
    org = [origin[0], origin[1], float(rng.choice([0.0, 0.3, -1.1, 2.4]))]
    layout = int(rng.choice([0, 3]))
    s = amd.BatchSim(num_envs=1, num_agents=1, num_beams=B, fov=fov, eps=eps, theta_dis=theta_dis, max_range=max_range, map_layout=layout)
    s.set_map_image(img, res, org)
    so = orc.ScanOracle(B, fov, eps=eps, theta_dis=theta_dis, max_range=max_range); so.set_map_dt(dt, res, org)
    kk = 60
    uu, vv = rng.uniform(-0.1, 1.1, kk) * W * res, rng.uniform(-0.1, 1.1, kk) * H * res
    c_, s_ = np.cos(org[2]), np.sin(org[2])
    poses = np.stack([org[0] + c_ * uu - s_ * vv, org[1] + s_ * uu + c_ * vv, rng.uniform(-20, 20, kk)], axis=1)
    ranges, hits, lk = s.scan_batch(poses, want_hits=True, want_lookups=True)
    nb = 0
    for i in range(kk):
        r, h_ = so.scan(poses[i], want_hits=True)
        if not (np.array_equal(ranges[i], r) and np.array_equal(hits[i], h_) and lk[i] == so.last_lookups):
            nb += 1
    print("scan ctor", mapname, "B", B, "fov", fov, "eps", eps, "theta_dis", theta_dis, "max_range", max_range, "yaw", org[2], "layout", layout, "bad poses", nb, "of", kk)
    s.close()
u.close()
