"""the CPU baseline's thread scaling on this box: oracle rollout (bench workload, 2048 envs x 2) at 1 .. all threads, and what
the container allows (cgroup cpu.max, affinity)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from _util import oracle_map_dt, bench_start_poses
from oracle import orc
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, "=", open(f).read().strip())
    except OSError:
        pass
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
dt, res, origin = oracle_map_dt("example_map")
E, A = 2048, 2
poses = bench_start_poses(E, A)
rng = np.random.default_rng(1000)
sets = np.stack([np.stack([rng.uniform(-0.2, 0.2, E * A), rng.uniform(2.0, 6.0, E * A)], axis=1) for _ in range(4)])
for native in (False, True):
    sim = orc.SimOracle(E, A, native=native); sim.set_map_dt(dt, res, origin)
    sim.set_noise(np.random.default_rng(12345).normal(0., .01, size=(200, 1080)))
    for th in (1, 8, 16, 32, 64, 128, 256):
        if th > (os.cpu_count() or 1):
            break
        sim.reset(poses)
        steps = 2 if th == 1 else (10 if th <= 16 else 40)
        sim.rollout(sets, 2, 20, poses, True, th)
        sim.reset(poses)
        t0 = time.perf_counter(); c0 = time.process_time()
        sim.rollout(sets, steps, 20, poses, True, th)
        el = time.perf_counter() - t0; cpu = time.process_time() - c0
        print("%s build  threads %3d  %9.0f agent-steps/s   wall %.2f s  cpu %.1f s  (busy cores %.1f)" % ("native" if native else "portable", th, E * A * steps / el, el, cpu, cpu / el))
