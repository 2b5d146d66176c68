"""a plain F110Env(num_agents=2) loop (BASELINE configs[0] through the HIP path) to put under rocprofv3: f110env_loop.py steps"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import f1tenth_gym_amd as amd
from _util import MAPS, bench_start_poses
n = int(sys.argv[1])
env = amd.F110Env(map=os.path.join(MAPS, "example_map"), map_ext=".png", num_agents=2)
env.reset(bench_start_poses(1, 2).reshape(2, 3))
act = np.array([[0.05, 3.0], [-0.05, 2.5]])
for _ in range(100):
    env.step(act)
t0 = time.perf_counter()
for _ in range(n):
    env.step(act)
dt = (time.perf_counter() - t0) / n
c, enq, wait = env.sim.batch.step_host_stats()
print("F110Env 1 env x 2 agents  %.1f us/step  in f110_step_host: enqueue %.1f us + wait %.1f us  (kernel launches per step: %s)"
      % (dt * 1e6, enq, wait, {1: "1 = k_step_tiny", 0: "the per-kernel form"}[env.sim.batch.step_launches()]))
