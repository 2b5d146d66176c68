"""A long auto-reset training-style loop: does device (and host) memory stay flat?
    gpurun -- 'python tools/debug/long_run_memory.py 1000000 > gpurun_out/long_run_memory.txt'
F110VecEnv(device_logic=True, auto_reset=True), scan noise drawn on the device from a deliberately
small row cache (64 rows), so episodes longer than the cache continue from the carried stream state."""
import os, sys, time, resource
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
import f1tenth_gym_amd as amd
from _util import MAPS, bench_start_poses
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 16
env = amd.F110VecEnv(E, auto_reset=True, device_logic=True, obs_fields=("poses_x", "poses_y", "collisions"),
                     map=os.path.join(MAPS, "example_map"), map_ext=".png", num_agents=2)
env.sim.batch.set_noise_rng(12345, 0.01, cache_rows=64)
poses = bench_start_poses(E, 2).reshape(E, 2, 3)
env.reset(poses)
rng = np.random.default_rng(0)
b = env.sim.batch
t0 = time.time(); n_done = 0; longest = 0
free0 = None
for t in range(steps):
    if t % 50 == 0:
        act = np.stack([rng.uniform(-0.15, 0.15, (E, 2)), rng.uniform(0.5, 4.0, (E, 2))], axis=2)
    obs, r, done, info = env.step(act)
    n_done += int(done.sum())
    if t % max(1, steps // 10) == 0 or t == steps - 1:
        free, total = b.device_mem_info()
        sc = b.get("step_count")["step_count"]
        longest = max(longest, int(sc.max()))
        if t >= steps // 10 and free0 is None:
            free0 = free
        print("step %8d  device free %.3f MiB  host maxrss %.1f MiB  episodes ended %d  longest live episode %d steps  %.0f steps/s"
              % (t, free / 2 ** 20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, n_done, longest, (t + 1) / (time.time() - t0)))
        sys.stdout.flush()
free, _ = b.device_mem_info()
print("device memory change since 10 %% of the run: %d bytes" % (free0 - free))
