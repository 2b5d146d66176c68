"""Randomised fuzzing of the episode bookkeeping on the device (lap toggles, lap counts and times, done, auto-reset re-seats:
f110_env.py:219-262 and :264-306, done by k_finalize_* when F110VecEnv(device_logic=True)) against the same bookkeeping in NumPy
on the host (`_LapLogic`, which tests/test_reference_fuzz.py pins to the LIVE reference F110Env): two F110VecEnv of one random
configuration — tracks, 1-4 cars, ego index, time step, integrator, beams, auto-reset, partial resets — driven with the same
actions, half of the envs in tight circles so that laps are completed.  Both run the same HIP step, so everything must be EQUAL.
    gpurun -- 'python tools/debug/fuzz_episode.py 0 40'      # seeds 0..39
"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _util import MAPS, oracle_map_dt, load_map_image
import f1tenth_gym_amd as amd

TRACKS = ["example_map", "berlin", "skirk"]
OBS = ("poses_x", "poses_y", "poses_theta", "linear_vels_x", "ang_vels_z", "collisions", "lap_counts", "scans")


def run(seed, verbose=True):
    rng = np.random.default_rng(200000 + seed)
    track = str(rng.choice(TRACKS)); E = int(rng.integers(2, 20)); A = int(rng.choice([1, 2, 2, 3, 4])); ego = int(rng.integers(0, A))
    time_step = float(rng.choice([0.01, 0.01, 0.02])); integ = [amd.Integrator.RK4, amd.Integrator.Euler][int(rng.random() < 0.3)]
    B = int(rng.choice([1080, 64, 271])); auto = bool(rng.random() < 0.6); T = int(rng.integers(150, 330))
    kw = dict(map=os.path.join(MAPS, track), map_ext='.png', num_agents=A, ego_idx=ego, timestep=time_step, integrator=integ,
              num_beams=B, seed=int(rng.integers(0, 1 << 30)), lidar_dist=float(rng.choice([0.0, 0.275])))
    tag = "seed %d %s E%d A%d ego%d dt%g %s B%d auto%d T%d" % (seed, track, E, A, ego, time_step, integ.name, B, auto, T)
    host = amd.F110VecEnv(E, auto_reset=auto, **kw)
    dev = amd.F110VecEnv(E, auto_reset=auto, device_logic=True, copy_obs=True, **kw)
    dt, res, origin = oracle_map_dt(track)
    rr, cc = np.nonzero((dt > 0.9) & (dt < 1.6))          # room for a circle of radius ~0.8 m
    poses = np.zeros((E, A, 3))
    for e in range(E):
        k = rng.integers(0, rr.shape[0])
        p0 = np.array([origin[0] + (cc[k] + 0.5) * res, origin[1] + (rr[k] + 0.5) * res, rng.uniform(0, 2 * np.pi)])
        for a in range(A):      # the other cars: beside the first, one car width apart (or on top of it, now and then)
            off = a * (0.55 if rng.random() < 0.9 else 0.1)
            poses[e, a] = [p0[0] - np.sin(p0[2]) * off, p0[1] + np.cos(p0[2]) * off, p0[2]]
    circles = rng.random(E) < 0.5

    def compare(t, h, d):
        (oh, rh, dh, ih), (od, rd, dd, idv) = h, d
        bad = []
        if not np.array_equal(dh, dd):
            bad.append("done")
        for k in OBS:
            if not np.array_equal(np.asarray(oh[k]), np.asarray(od[k])):
                bad.append(k)
        if not np.allclose(oh["lap_times"], od["lap_times"], rtol=0, atol=1e-12):
            bad.append("lap_times")
        for k in ("checkpoint_done", "toggle_list", "near_starts"):
            if k in ih and k in idv and not np.array_equal(np.asarray(ih[k]), np.asarray(idv[k])):
                bad.append(k)
        if bad:
            print("MISMATCH", tag, "step", t, bad)
            if verbose:
                for k in bad[:3]:
                    a_, b_ = (dh, dd) if k == "done" else (oh.get(k, ih.get(k)), od.get(k, idv.get(k)))
                    print("   ", k, "host", np.asarray(a_).ravel()[:12], "device", np.asarray(b_).ravel()[:12])
        return not bad

    ok = compare(-1, host.reset(poses), dev.reset(poses))
    dones = laps = 0
    for t in range(T):
        if not ok:
            break
        if t % 25 == 0:
            act = np.stack([rng.uniform(-0.3, 0.3, (E, A)), rng.uniform(0.5, 5.0, (E, A))], axis=2)
            act[circles, :, 0] = 0.4; act[circles, :, 1] = rng.uniform(2.0, 3.5)
        h, d = host.step(act), dev.step(act)
        ok = compare(t, h, d)
        dones += int(np.sum(h[2])); laps = max(laps, int(np.max(h[0]["lap_counts"])))
        if ok and rng.random() < 0.03:
            mask = rng.random(E) < 0.3
            if not mask.all():
                ok = compare(t, host.reset(poses, mask), dev.reset(poses, mask))
    host.sim.batch.close(); dev.sim.batch.close()
    if ok:
        print("ok", tag, "done flags %d, most laps %d" % (dones, laps))
    return ok


if __name__ == "__main__":
    a, b = int(sys.argv[1]), int(sys.argv[2])
    bad = [sd for sd in range(a, b) if not run(sd)]
    print("failed seeds:", bad)
