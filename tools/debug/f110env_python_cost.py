"""What F110Env.step costs in Python AROUND the library call: the step with f110_step_host replaced by nothing (the observation block keeps
the last real step's contents), and the parts of it one by one.    python tools/debug/f110env_python_cost.py [agents=2]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import f1tenth_gym_amd as amd
from f1tenth_gym_amd import workload

A = int(sys.argv[1]) if len(sys.argv) > 1 else 2
env = amd.F110Env(map=workload.map_stem("example_map"), map_ext=".png", num_agents=A)
env.reset(workload.bench_start_poses(1, A).reshape(A, 3))
act = np.array([[0.05, 3.0], [-0.05, 2.5]])[:A]
for _ in range(300):
    env.step(act)


def timed(fn, n=40000):
    for _ in range(2000):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


b = env.sim._b
real = b.step_host_inplace
b.step_host_inplace = lambda hb: None
print("F110Env.step without the library call   %.2f us" % timed(lambda: env.step(act)))
print("  Simulator.step without the call       %.2f us" % timed(lambda: env.sim.step(act)))
obs = env.sim.step(act)
lap = env._lap
print("  lap bookkeeping (update_single)       %.2f us" % timed(lambda: lap.update_single(obs['poses_x'], obs['poses_y'], obs['collisions'], 0.01)))
hb = env.sim._hb
v = hb.views
print("  np.asarray(action).reshape            %.2f us" % timed(lambda: np.asarray(act, dtype=np.float64).reshape(A, 2)))
print("  hb.actions[...] = actions             %.2f us" % timed(lambda: hb.actions.__setitem__(Ellipsis, act)))
print("  six .copy() of the block's views      %.2f us" % timed(lambda: (v["scans"].copy(), v["state"].copy(), v["agent_poses"].T.copy(), v["collisions"].copy(), v["collision_idx"].copy(), v["in_collision"].copy())))
print("  scans.copy() alone                    %.2f us" % timed(lambda: v["scans"].copy()))
b.step_host_inplace = real
print("  the ctypes call on a closed-over no-op: n/a; real call incl. GPU  %.2f us" % timed(lambda: real(hb), 5000))
env.sim.batch.close()
