"""Randomised parity fuzzing of the HOST-BLOCK step (f110_step_host: what F110Env.step / F110VecEnv.step call) against the CPU oracle,
weighted towards TINY batches — one or two envs of one or two cars, where the whole step is ONE launch (k_step_tiny: integration per scan
wave, shadow columns, last-workgroup finalize + observation block + completion word) — next to batches that take the pair kernel's fused
epilogue (2 cars per env, more than 4 agents) and k_host_block (1 or 3 cars per env).  Every column of the page-locked block is compared
with the oracle after every step: flags / indices exact, floats <= 1e-9; the NumPy noise stream comes from the device generator on one
side and from NumPy on the other; masked re-seats in between.
    gpurun -- 'python tools/debug/fuzz_host.py 0 40'      # seeds 0..39
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
    rr, cc = np.nonzero((dt > lo) & (dt < hi))
    k = rng.integers(0, rr.shape[0], n)
    mx, my = (cc[k] + 0.5) * res, (rr[k] + 0.5) * res
    c, s = np.cos(origin[2]), np.sin(origin[2])
    return np.stack([origin[0] + c * mx - s * my, origin[1] + s * mx + c * my, rng.uniform(0.0, 2 * np.pi, n)], axis=1)


def run(seed, verbose=True, tol=1e-9):
    rng = np.random.default_rng(700000 + seed)
    E, A = [(1, 2), (1, 1), (2, 2), (2, 1), (4, 1), (3, 1), (1, 2), (3, 2), (9, 2), (5, 1), (2, 3), (40, 2)][int(rng.integers(0, 12))]
    B = int(rng.choice([1080, 1080, 1080, 720, 271])); fov = float(rng.choice([4.7, 4.7, 6.0]))
    integ = int(rng.choice([1, 1, 2])); ld = float(rng.choice([0.0, 0.275])); time_step = float(rng.choice([0.01, 0.01, 0.02]))
    eps = float(rng.choice([1e-4, 1e-4, 0.03])); theta_dis = int(rng.choice([2000, 2000, 3600])); max_range = float(rng.choice([30.0, 30.0, 10.0]))
    T = int(rng.integers(60, 180)); noise_mode = str(rng.choice(["rng", "rng", "table", "off"])); nseed = int(rng.integers(0, 10 ** 6))
    track = str(rng.choice(TRACKS)); yaw = float(rng.choice([0.0, 0.0, 0.4, -2.0]))
    kw = dict(num_beams=B, fov=fov, integrator=integ, lidar_dist=ld, eps=eps, theta_dis=theta_dis, max_range=max_range, time_step=time_step)
    tag = "seed %d E%d A%d B%d fov%.2f integ%d ld%.3f T%d noise %s eps%g td%d mr%g dt%g %s yaw%g" % (seed, E, A, B, fov, integ, ld, T, noise_mode, eps, theta_dis,
                                                                                                max_range, time_step, track, yaw)
    _, res, origin = load_map_image(track); dt, _, _ = oracle_map_dt(track)
    origin = [origin[0], origin[1], yaw]
    s = amd.BatchSim(num_envs=E, num_agents=A, **kw)
    s.set_map_dt(dt, res, origin)
    ref = orc.SimOracle(E, A, **kw); ref.set_map_dt(dt, res, origin)
    if noise_mode != "off":
        noise = np.random.default_rng(nseed).normal(0., 0.01, size=(T + 2, B))
        ref.set_noise(noise)
        if noise_mode == "rng":
            s.set_noise_rng(nseed, 0.01)       # the device draws NumPy's stream itself
        else:
            s.set_noise_table(noise)
    if rng.random() < 0.5 and A > 1:
        p2 = dict(amd.DEFAULT_PARAMS); p2.update({'mu': 0.8, 'length': 0.5, 'width': 0.27, 'v_max': 14.0, 'a_max': 8.0})
        s.set_params(p2, A - 1); ref.set_params(p2, A - 1)
    poses = np.zeros((E * A, 3))
    for e in range(E):
        p0 = free_poses(dt, res, origin, rng, 1)[0]
        for a in range(A):
            poses[e * A + a] = p0 + (0 if a == 0 else 1) * np.array([rng.uniform(-0.9, 0.9), rng.uniform(-0.9, 0.9), rng.uniform(-0.6, 0.6)])
    s.reset(poses); ref.reset(poses)
    fields = ("scans", "state", "collisions", "collision_idx", "in_collision", "agent_poses")
    hb = s.host_block(fields)
    # which form the library must pick (f110_hip.hip tiny_applies): a waiting host step of <= 4 agents, 1-2 per env, agent-aligned scan
    lanes = (B + 63) // 64 * 64
    want_tiny = int(E * A <= 4 and A <= 2 and (lanes - B) * 100 <= 3 * B)
    hits = 0
    for t in range(T):
        if t % 9 == 0:
            act = np.stack([rng.uniform(-0.3, 0.3, E * A), rng.uniform(0.5, 7.0, E * A)], axis=1)
        hb.actions[...] = act
        s.step_host(hb)
        ref.step(act)
        if s.step_launches() != want_tiny:
            print("MISMATCH", tag, "step", t, "launch form", s.step_launches(), "expected", want_tiny)
            s.close()
            return False
        v = hb.views
        got = {"scans": v["scans"], "state": v["state"].T, "collisions": v["collisions"], "collision_idx": v["collision_idx"],
               "in_collision": v["in_collision"], "agent_poses": v["agent_poses"].T}
        if np.abs(ref.state).max() > 1e6:
            tag += " (stopped at step %d: dynamics diverged)" % t
            break
        hits += int(ref.collisions.sum())
        flags = (np.array_equal(got["collisions"], ref.collisions) and np.array_equal(got["in_collision"], ref.in_collision)
                 and np.array_equal(got["collision_idx"], ref.collision_idx))
        es, er, ep = rel_err(got["state"], ref.state), rel_err(got["scans"], ref.scans), rel_err(got["agent_poses"], ref.agent_poses)
        o = s.get("state", "scans", "step_count")     # ... and the device-side columns behind the block
        dev_ok = np.array_equal(o["state"], got["state"]) and np.array_equal(o["scans"], got["scans"]) and np.array_equal(o["step_count"], ref.step_count)
        if not flags or not es < tol or not er < tol or not ep < tol or not dev_ok:
            print("MISMATCH", tag, "step", t, "flags", flags, "state", es, "scan", er, "poses", ep, "device == block", dev_ok)
            if verbose:
                print("   gpu state", got["state"], "\n   ref state", ref.state, "\n   collisions", got["collisions"], ref.collisions)
            s.close()
            return False
        if rng.random() < 0.06:
            mask = (rng.random(E) < 0.5).astype(np.uint8)
            s.reset(poses, mask); ref.reset(poses, mask)
    s.close()
    print("ok", tag, "one launch per step" if want_tiny else "per-kernel form", "| collision flags seen %d of %d" % (hits, T * E * A))
    return True


if __name__ == "__main__":
    a, b = int(sys.argv[1]), int(sys.argv[2])
    bad = [sd for sd in range(a, b) if not run(sd)]
    print("failed seeds:", bad)
