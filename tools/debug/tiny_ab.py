"""k_step_tiny (one launch per step, lab switch step_tiny = 1, the default) against the three-kernel form (step_tiny = 0) on the shapes
it serves — the experimental build carries the switch:
    F110_LIB_VARIANT=experimental python tools/debug/tiny_ab.py
(i) F110Env(num_agents=2).step, one env: the reference's own loop; (ii) F110VecEnv(E, device_logic=True).step, host actions in, done out;
(iii) BatchSim.step_device back to back, one sync at the end (a device-resident loop)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import f1tenth_gym_amd as amd
from f1tenth_gym_amd import workload

MAP = workload.map_stem("example_map")


def timed(fn, n, warm=200):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


for rep in range(2):
    for tiny in (1, 0):
        os.environ["F110_EXP"] = "step_tiny=%d" % tiny
        env = amd.F110Env(map=MAP, map_ext=".png", num_agents=2)
        env.reset(workload.bench_start_poses(1, 2).reshape(2, 3))
        act = np.array([[0.05, 3.0], [-0.05, 2.5]])
        us = timed(lambda: env.step(act), 3000)
        _, enq, wait = env.sim.batch.step_host_stats()
        line = "step_tiny %d | F110Env 2 cars %.1f us (enqueue %.1f wait %.1f)" % (tiny, us, enq, wait)
        env.sim.batch.close()
        env = amd.F110Env(map=MAP, map_ext=".png", num_agents=1)
        env.reset(workload.bench_start_poses(1, 1).reshape(1, 3))
        us = timed(lambda: env.step(np.array([[0.05, 3.0]])), 3000)
        line += " | F110Env 1 car %.1f us" % us
        env.sim.batch.close()
        for E in (8, 32):
            v = amd.F110VecEnv(E, map=MAP, map_ext=".png", num_agents=2, auto_reset=True, device_logic=True, obs_fields=())
            v.reset(workload.bench_start_poses(E, 2).reshape(E, 2, 3))
            a = np.tile([0.05, 3.0], (E, 2, 1))
            us = timed(lambda: v.step(a), 2000)
            line += " | VecEnv %d envs %.1f us" % (E, us)
            v.sim.batch.close()
        for E in (1, 32):
            b = amd.BatchSim(num_envs=E, num_agents=2)
            b.set_map(MAP + ".yaml", ".png"); b.set_noise_rng(12345, 0.01)
            b.reset(workload.bench_start_poses(E, 2))
            d = b.device_array((2 * E, 2)); d.upload(np.tile([0.05, 3.0], (2 * E, 1)))
            for _ in range(200):
                b.step_device(d)
            b.sync()
            t0 = time.perf_counter()
            for _ in range(2000):
                b.step_device(d)
            b.sync()
            line += " | device loop %d envs %.1f us" % (E, (time.perf_counter() - t0) / 2000 * 1e6)
            b.close()
        print(line, flush=True)
del os.environ["F110_EXP"]
