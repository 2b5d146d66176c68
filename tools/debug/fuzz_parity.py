"""Randomised step-level parity fuzzing: HIP path vs the CPU oracle over random configurations.
    gpurun -- 'python tools/debug/fuzz_parity.py 0 40'      # seeds 0..39
Prints one line per seed and details of the first mismatch."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _util import load_map_image, oracle_map_dt, raceline
from oracle import orc
import f1tenth_gym_amd as amd


def run(seed, verbose=True, tol=1e-9, stop_when_diverged=False):
    rng = np.random.default_rng(seed)
    mapname = rng.choice(["example_map", "berlin", "skirk"])
    A = int(rng.choice([1, 2, 2, 3, 4])); E = int(rng.integers(1, 40))
    B = int(rng.choice([1080, 1080, 64, 100, 271, 720, 1500, 2500])); fov = float(rng.choice([4.7, 4.7, 6.0, 3.0, 6.28]))
    integ = int(rng.choice([1, 1, 2])); ld = float(rng.choice([0.0, 0.0, 0.275])); layout = int(rng.integers(0, 4))
    layout = {1: 0, 2: 3}.get(layout, layout)   # (the random draw is kept as it was: layouts 1 / 2 — tiles, byte codes — were retired in round 5)
    T = int(rng.integers(20, 70)); nrows = int(rng.choice([0, 5, T + 2]))
    tasks = int(rng.choice([0, 1, 3])); block = int(rng.choice([0, 64, 128, 256]))
    img, res, origin = load_map_image(mapname); dt, _, _ = oracle_map_dt(mapname)
    # round 5: the ScanSimulator2D constructor arguments and the yaw of the map origin join the draw — from a generator of their
    # own, so that a seed's other choices stay what they were (laser_models.py:360-381: eps, theta_dis, max_range; :417-420: origin[2])
    rng2 = np.random.default_rng(seed + 7777)
    eps = float(rng2.choice([1e-4, 1e-4, 1e-4, 0.03, 0.2])); theta_dis = int(rng2.choice([2000, 2000, 2000, 720, 1000, 3600]))
    max_range = float(rng2.choice([30.0, 30.0, 30.0, 8.0, 12.5])); yaw = float(rng2.choice([0.0, 0.0, 0.3, -1.1, 2.4]))
    origin = [origin[0], origin[1], yaw]
    time_step = float(rng2.choice([0.01, 0.01, 0.005, 0.02]))
    if rng2.random() < 0.12:      # a batch inside the longest-first window of the scan (12 000+ tasks), a few steps of it
        E = int(rng2.integers(12000 // (A * ((B + 63) // 64)) + 1, 12000 // (A * ((B + 63) // 64)) + 400)); T = int(rng2.integers(5, 10))
    s = amd.BatchSim(num_envs=E, num_agents=A, num_beams=B, fov=fov, integrator=integ, lidar_dist=ld, map_layout=layout,
                     scan_tasks_per_wave=tasks, scan_block=block, eps=eps, theta_dis=theta_dis, max_range=max_range, time_step=time_step)
    s.set_map_image(img, res, origin)
    ref = orc.SimOracle(E, A, num_beams=B, fov=fov, integrator=integ, lidar_dist=ld, eps=eps, theta_dis=theta_dis, max_range=max_range, time_step=time_step)
    ref.set_map_dt(dt, res, origin)
    # the oracle builds its beam tables with libm; use the product's NumPy tables on both sides
    if nrows:
        noise = np.random.default_rng(seed + 1).normal(0., 0.01, size=(nrows, B))
        s.set_noise_table(noise); ref.set_noise(noise)
    if rng.random() < 0.4 and A > 1:
        p2 = dict(amd.DEFAULT_PARAMS); p2.update({'mu': 0.7, 'length': 0.45, 'width': 0.25, 'v_max': 12.0})
        s.set_params(p2, A - 1); ref.set_params(p2, A - 1)
    if mapname == "example_map":
        w = raceline(); k = rng.integers(0, w.shape[0], E)
        base = np.stack([w[k, 1], w[k, 2], w[k, 3] + np.pi / 2], axis=1)
    else:
        base = np.stack([rng.uniform(-0.5, 0.5, E), rng.uniform(-0.5, 0.5, E), rng.uniform(0, 6.28, E)], axis=1)
    poses = np.repeat(base, A, axis=0) + np.stack([rng.uniform(-0.8, 0.8, E * A), rng.uniform(-0.8, 0.8, E * A), rng.uniform(-0.5, 0.5, E * A)], axis=1)
    if yaw != 0.0:   # the map turns about its origin: the cars turn with it
        c_, s_ = np.cos(yaw), np.sin(yaw)
        dx, dy = poses[:, 0] - origin[0], poses[:, 1] - origin[1]
        poses = np.stack([origin[0] + c_ * dx - s_ * dy, origin[1] + s_ * dx + c_ * dy, poses[:, 2] + yaw], axis=1)
    s.reset(poses); ref.reset(poses)
    tag = "seed %d %s E%d A%d B%d fov%.2f integ%d ld%.3f layout%d T%d noise%d tasks%d blk%d eps%g td%d mr%g yaw%g dt%g" % (
        seed, mapname, E, A, B, fov, integ, ld, layout, T, nrows, tasks, block, eps, theta_dis, max_range, yaw, time_step)
    for t in range(T):
        if t % 7 == 0:
            act = np.stack([rng.uniform(-0.45, 0.45, E * A), rng.uniform(-4.0, 9.0, E * A)], axis=1)
        s.step(act); ref.step(act, 8)
        if rng.random() < 0.1:
            mask = (rng.random(E) < 0.3).astype(np.uint8)
            s.reset(poses, mask); ref.reset(poses, mask)
        o = s.get("scans", "state", "collisions", "collision_idx", "in_collision", "step_count")
        big = np.abs(ref.state).max() > 1e6
        if big and stop_when_diverged:
            # the random actions blew a car's yaw rate / slip past 1e6: from here on 1-ulp libm differences
            # are amplified without bound (and hidden again by the next wall hit's zeroing) — every step
            # up to this one has been compared
            tag += " (stopped at step %d: dynamics diverged)" % t
            break
        bad_flags = int(np.sum(o["collisions"] != ref.collisions) + np.sum(o["in_collision"] != ref.in_collision) + np.sum(o["collision_idx"] != ref.collision_idx))
        def rel(a, b):   # |a - b| <= tol*|b| + 1e-12 per element (tests/_util.rel_err), element-wise
            with np.errstate(invalid="ignore", divide="ignore"):
                ex = np.maximum(np.abs(a - b) - 1e-12, 0.0)
                return np.where(ex > 0.0, ex / np.abs(b), np.where(np.isnan(ex), np.nan, 0.0))
        es = np.max(rel(o["state"], ref.state))
        dsc = rel(o["scans"], ref.scans)
        dsc = np.where(np.isnan(dsc), np.where(np.isnan(o["scans"]) == np.isnan(ref.scans), 0.0, np.inf), dsc)
        er = dsc.max()
        if bad_flags or es > tol or er > tol or not np.array_equal(o["step_count"], ref.step_count):
            print("MISMATCH", tag, "step", t, "flags", bad_flags, "state", es, "scan", er, "diverged" if big else "")
            if verbose:
                i, b = np.unravel_index(np.argmax(dsc), dsc.shape)
                print("   agent", i, "beam", b, "gpu", o["scans"][i, b], "ref", ref.scans[i, b], "state", ref.state[i], "wall", ref.in_collision[i])
            s.close()
            return False
    s.close()
    print("ok", tag)
    return True


if __name__ == "__main__":
    a, b = int(sys.argv[1]), int(sys.argv[2])
    # (as the suite runs it: a seed whose random actions blow a car's state past 1e6 is compared up to that step)
    bad = [sd for sd in range(a, b) if not run(sd, stop_when_diverged=True)]
    print("failed seeds:", bad)
