"""Randomised parity fuzzing of what fuzz_parity.py leaves fixed: a different track per env (f110_add_map_dt /
f110_set_env_maps), a vehicle parameter set per agent (f110_set_params_batch), and actions calm enough that a rollout runs its
whole length (wall hits and car-to-car hits included) instead of stopping when the dynamics blow up.  The HIP step is
compared with one CPU oracle per env (that env's track, that env's parameter sets).
    gpurun -- 'python tools/debug/fuzz_envs.py 0 40'      # seeds 0..39
"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _util import load_map_image, oracle_map_dt, rel_err
from oracle import orc
import f1tenth_gym_amd as amd

TRACKS = ["example_map", "berlin", "skirk"]


def free_poses(dt, res, origin, rng, n, lo=0.35, hi=1.5):
    """n poses on cells whose wall distance is in (lo, hi): inside the track, not in the free space around it"""
    rr, cc = np.nonzero((dt > lo) & (dt < hi))
    k = rng.integers(0, rr.shape[0], n)
    mx, my = (cc[k] + 0.5) * res, (rr[k] + 0.5) * res
    c, s = np.cos(origin[2]), np.sin(origin[2])
    return np.stack([origin[0] + c * mx - s * my, origin[1] + s * mx + c * my, rng.uniform(0.0, 2 * np.pi, n)], axis=1)


def run(seed, verbose=True, tol=1e-9):
    rng = np.random.default_rng(100000 + seed)
    E = int(rng.integers(2, 14)); A = int(rng.choice([1, 2, 2, 3, 4])); K = int(rng.integers(1, 4))
    B = int(rng.choice([1080, 1080, 64, 271, 720, 1500])); fov = float(rng.choice([4.7, 4.7, 6.0, 3.0]))
    integ = int(rng.choice([1, 1, 2])); ld = float(rng.choice([0.0, 0.275])); layout = int(rng.choice([0, 3]))
    eps = float(rng.choice([1e-4, 1e-4, 0.03])); theta_dis = int(rng.choice([2000, 2000, 720, 3600]))
    max_range = float(rng.choice([30.0, 30.0, 10.0])); time_step = float(rng.choice([0.01, 0.01, 0.02]))
    T = int(rng.integers(60, 160)); nrows = int(rng.choice([0, 7, T + 2]))
    per_agent = rng.random() < 0.6
    tracks = [str(t) for t in rng.choice(TRACKS, K, replace=False)]
    if K > 1:
        layout = 3      # f110_set_env_maps: the padded layout only
    yaws = [float(rng.choice([0.0, 0.0, 0.4, -2.0])) for _ in tracks]
    env_map = rng.integers(0, K, E) if rng.random() < 0.7 else (np.arange(E) * K) // E
    tag = "seed %d E%d A%d B%d fov%.2f integ%d ld%.3f layout%d T%d noise%d eps%g td%d mr%g dt%g tracks %s yaws %s env_map %s params %s" % (
        seed, E, A, B, fov, integ, ld, layout, T, nrows, eps, theta_dis, max_range, time_step, ",".join(tracks), yaws,
        "".join(str(int(m)) for m in env_map), "per-agent" if per_agent else "per-slot")
    kw = dict(num_beams=B, fov=fov, integrator=integ, lidar_dist=ld, eps=eps, theta_dis=theta_dis, max_range=max_range, time_step=time_step)
    s = amd.BatchSim(num_envs=E, num_agents=A, map_layout=layout, **kw)
    maps = []
    for k, (name, yaw) in enumerate(zip(tracks, yaws)):
        _, res, origin = load_map_image(name); dt, _, _ = oracle_map_dt(name)
        origin = [origin[0], origin[1], yaw]
        slot = 0
        if k == 0:
            s.set_map_dt(dt, res, origin)
        else:
            slot = s.add_map_dt(dt, res, origin)
        assert slot == k
        maps.append((dt, res, origin))
    if K > 1:
        s.set_env_maps(env_map)
    noise = None
    if nrows:
        noise = np.random.default_rng(seed + 1).normal(0., 0.01, size=(nrows, B)); s.set_noise_table(noise)

    def draw_params():
        p = dict(amd.DEFAULT_PARAMS)
        p.update({'mu': rng.uniform(0.6, 1.2), 'm': rng.uniform(3.0, 4.2), 'lf': rng.uniform(0.147, 0.17), 'C_Sf': rng.uniform(4.0, 5.5),
                  'a_max': rng.uniform(7.0, 10.0), 'v_max': rng.uniform(12.0, 22.0), 'length': rng.uniform(0.5, 0.62),
                  'width': rng.uniform(0.27, 0.34), 'sv_max': rng.uniform(2.5, 3.4), 'v_switch': rng.uniform(6.0, 8.0)})
        return p
    if per_agent:
        sets = [draw_params() for _ in range(E * A)]
        s.set_params_batch(sets)
    else:
        slot_sets = [draw_params() if rng.random() < 0.5 else dict(amd.DEFAULT_PARAMS) for _ in range(A)]
        for a in range(A):
            s.set_params(slot_sets[a], a)
        sets = [slot_sets[i % A] for i in range(E * A)]
    # cars of one env start near each other (so that they meet), on that env's track
    poses = np.zeros((E * A, 3))
    for e in range(E):
        dt, res, origin = maps[int(env_map[e])]
        p0 = free_poses(dt, res, origin, rng, 1)[0]
        cand = free_poses(dt, res, origin, rng, 4000)
        cand = cand[np.hypot(cand[:, 0] - p0[0], cand[:, 1] - p0[1]) < 3.0]
        placed = [p0]
        for q in cand:      # the other cars: on the track within 3 m of the first, most of them clear of each other
            if len(placed) == A:
                break
            if min(np.hypot(q[0] - w[0], q[1] - w[1]) for w in placed) > (0.75 if rng.random() < 0.85 else 0.3):
                placed.append(np.array([q[0], q[1], p0[2] + rng.uniform(-0.6, 0.6)]))
        while len(placed) < A:
            placed.append(p0 + np.array([rng.uniform(-0.9, 0.9), rng.uniform(-0.9, 0.9), rng.uniform(-0.6, 0.6)]))
        poses[e * A:(e + 1) * A] = np.stack(placed)
    refs = []
    for e in range(E):
        r = orc.SimOracle(1, A, **kw)
        dt, res, origin = maps[int(env_map[e])]
        r.set_map_dt(dt, res, origin)
        if noise is not None:
            r.set_noise(noise)
        for a in range(A):
            r.set_params(sets[e * A + a], a)
        r.reset(poses[e * A:(e + 1) * A]); refs.append(r)
    s.reset(poses)
    hits = 0
    for t in range(T):
        if t % 9 == 0:
            act = np.stack([rng.uniform(-0.25, 0.25, E * A), rng.uniform(0.5, 6.0, E * A)], axis=1)
        s.step(act)
        for e in range(E):
            refs[e].step(act[e * A:(e + 1) * A])
        if rng.random() < 0.05:
            mask = (rng.random(E) < 0.4).astype(np.uint8)
            s.reset(poses, mask)
            for e in np.nonzero(mask)[0]:
                refs[e].reset(poses[e * A:(e + 1) * A])
        o = s.get("scans", "state", "collisions", "collision_idx", "in_collision", "step_count")
        if max(np.abs(r.state).max() for r in refs) > 1e6:   # (as fuzz_parity.py: past this, 1-ulp differences are amplified without bound)
            tag += " (stopped at step %d: dynamics diverged)" % t
            break
        for e in range(E):
            r, sl = refs[e], slice(e * A, (e + 1) * A)
            hits += int(r.collisions.sum())
            flags = (np.array_equal(o["collisions"][sl], r.collisions) and np.array_equal(o["in_collision"][sl], r.in_collision)
                     and np.array_equal(o["collision_idx"][sl], r.collision_idx) and np.array_equal(o["step_count"][sl], r.step_count))
            es, er = rel_err(o["state"][sl], r.state), rel_err(o["scans"][sl], r.scans)
            if not flags or not es < tol or not er < tol:
                print("MISMATCH", tag, "step", t, "env", e, "flags", flags, "state", es, "scan", er)
                if verbose:
                    print("   gpu state", o["state"][sl], "\n   ref state", r.state, "\n   collisions", o["collisions"][sl], r.collisions)
                s.close()
                return False
    s.close()
    print("ok", tag, "collision flags seen %d of %d" % (hits, T * E * A))
    return True


if __name__ == "__main__":
    a, b = int(sys.argv[1]), int(sys.argv[2])
    bad = [sd for sd in range(a, b) if not run(sd)]
    print("failed seeds:", bad)
