"""steps/s of the RL-facing F110VecEnv (device episode logic, auto-reset), observations left in HBM"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import f1tenth_gym_amd as amd
from _util import MAPS, bench_start_poses
for E in (2048, 32768):
    for fields in ((), ("poses_x", "poses_y", "poses_theta", "collisions")):
        env = amd.F110VecEnv(E, auto_reset=True, device_logic=True, obs_fields=fields, map=os.path.join(MAPS, "example_map"), map_ext=".png", num_agents=2)
        poses = bench_start_poses(E, 2).reshape(E, 2, 3)
        env.reset(poses)
        rng = np.random.default_rng(0)
        act = np.stack([rng.uniform(-0.2, 0.2, (E, 2)), rng.uniform(2, 6, (E, 2))], axis=2)
        for _ in range(10): env.step(act)
        t0 = time.perf_counter(); n = 100
        for _ in range(n): env.step(act)
        dt = (time.perf_counter() - t0) / n
        print("E=%6d obs_fields=%-50s %.3f ms/step  %.1f M agent-steps/s" % (E, fields, dt * 1e3, E * 2 / dt / 1e6))
