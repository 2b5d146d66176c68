"""F110Env(num_agents = A) driven like the reference's users drive it — reset(), step() until done, reset() ... — against the CPU oracle at
EVERY step, with the device noise generator's row cache cut to 64 rows: an episode that outlives the cache leaves the one-launch form
(k_step_tiny) for the per-kernel form with k_noise_rows in MID-EPISODE, the next reset() brings the one-launch form back.  Prints one line.
    python tools/debug/f110env_soak.py [episodes=100] [agents=2] [cache rows=64] [longest episode=400] [slow=0]
cache rows 0 = the default cache (16 384 rows, generated AHEAD of need on their own stream in doublings from 256: noise_start_ahead); with
slow=1 the cars crawl, so that episodes live through several doublings."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _util import oracle_map_dt, rel_err, map_stem
from oracle import orc
import f1tenth_gym_amd as amd
from f1tenth_gym_amd import workload


def run(episodes=100, A=2, cap=400, verbose=True, rows=64, slow=False):
    dt, res, origin = oracle_map_dt("example_map")
    env = amd.F110Env(map=map_stem("example_map"), map_ext=".png", num_agents=A, seed=12345)
    if rows:
        env.sim.batch.set_noise_rng(12345, 0.01, cache_rows=rows)
    ref = orc.SimOracle(1, A); ref.set_map_dt(dt, res, origin)
    ref.set_noise(np.random.default_rng(12345).normal(0., 0.01, size=(cap + 3, 1080)))
    rng = np.random.default_rng(5)
    forms = {0: 0, 1: 0}
    steps = switches = 0
    worst = 0.0
    longest = 0
    for ep in range(episodes):
        poses = workload.start_poses(np.array([int(rng.integers(0, 783))]), A, gap_wp=int(rng.integers(3, 9)))
        obs, _, done, _ = env.reset(poses)
        ref.reset(poses); ref.step(np.zeros((A, 2)))          # f110_env.py:337-338: reset advances one zero-action step
        last = env.sim.batch.step_launches()
        t = 0
        while True:
            sc, st = np.stack(obs["scans"]), ref.state
            ok = (np.array_equal(np.asarray(obs["collisions"]), ref.collisions) and rel_err(sc, ref.scans) < 1e-9 and
                  rel_err(np.array(obs["poses_x"]), st[:, 0]) < 1e-9 and rel_err(np.array(obs["poses_theta"]), st[:, 4]) < 1e-9 and
                  rel_err(np.array(obs["linear_vels_x"]), st[:, 3]) < 1e-9)
            worst = max(worst, rel_err(sc, ref.scans))
            if not ok:
                print("MISMATCH episode", ep, "step", t, "form", last)
                return False
            if done or t >= cap:
                longest = max(longest, t)
                break
            if t % 15 == 0:
                act = np.stack([rng.uniform(-0.25, 0.25, A), rng.uniform(1.0, 6.0, A)], axis=1)
                if slow:
                    act = np.stack([rng.uniform(-0.08, 0.08, A), rng.uniform(0.2, 0.7, A)], axis=1)
            obs, _, done, _ = env.step(act)
            ref.step(act)
            form = env.sim.batch.step_launches()
            forms[form] += 1
            switches += int(form != last)
            last = form
            t += 1; steps += 1
    env.sim.batch.close()
    if verbose:
        print("F110Env x %d car(s): %d episodes, %d steps vs the oracle at every step: no mismatch (largest scan difference %.1e relative); %d steps as ONE launch, "
              "%d in the per-kernel form (row cache of %s outlived), %d switches between the forms in mid-episode / at reset; longest episode %d steps" % (A, episodes, steps, worst, forms[1], forms[0], rows or "16384 (default)", switches, longest))
    return forms, switches, longest


if __name__ == "__main__":
    arg = lambda k, d: int(sys.argv[k]) if len(sys.argv) > k else d
    ok = run(arg(1, 100), arg(2, 2), cap=arg(4, 400), rows=arg(3, 64), slow=bool(arg(5, 0)))
    sys.exit(0 if ok else 1)
