"""a plain F110VecEnv(device_logic=True, auto_reset=True) loop to put under rocprofv3: vecenv_loop.py E steps [spin]"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import f1tenth_gym_amd as amd
from _util import MAPS, bench_start_poses
E, n = int(sys.argv[1]), int(sys.argv[2])
env = amd.F110VecEnv(E, auto_reset=True, device_logic=True, obs_fields=(), spin_wait=len(sys.argv) > 3,
                     map=os.path.join(MAPS, "example_map"), map_ext=".png")
env.reset(bench_start_poses(E, 2).reshape(E, 2, 3))
rng = np.random.default_rng(0)
env.action_buffer[...] = np.stack([rng.uniform(-0.2, 0.2, (E, 2)), rng.uniform(2, 6, (E, 2))], axis=2)
for _ in range(50):
    env.step(None)
t0 = time.perf_counter()
for _ in range(n):
    env.step(None)
dt = (time.perf_counter() - t0) / n
print("E=%d  %.4f ms/step  %.2f M agent-steps/s  host stats %s" % (E, dt * 1e3, 2 * E / dt / 1e6, env.sim.batch.step_host_stats()))
