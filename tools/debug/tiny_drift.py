"""Does the host-synchronised tiny step's time depend on how long the loop has been running?  Per block of 1000 steps: seconds since the loop
started, in-call enqueue + wait.  Then a pause (the GPU idles) and again.    python tools/debug/tiny_drift.py [agents per env = 2] [blocks = 14]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import f1tenth_gym_amd as amd
from f1tenth_gym_amd import workload

A = int(sys.argv[1]) if len(sys.argv) > 1 else 2
BLOCKS = int(sys.argv[2]) if len(sys.argv) > 2 else 14
s = amd.BatchSim(num_envs=1, num_agents=A)
s.set_map(workload.map_stem("example_map") + ".yaml", ".png"); s.set_noise_rng(12345, 0.01)
poses = workload.bench_start_poses(1, A)
s.reset(poses)
hb = s.host_block(("scans", "state", "agent_poses", "collisions", "collision_idx", "in_collision"))
hb.actions[...] = np.tile([0.05, 3.0], (A, 1))
for phase, pause in (("from a cold start", 0.0), ("after a 0.2 s pause", 0.2), ("after a 2 s pause", 2.0)):
    time.sleep(pause)
    s.reset(poses)
    print("--", phase, flush=True)
    t0 = time.perf_counter()
    for b in range(BLOCKS):
        s.step_host_stats()
        for _ in range(1000):
            s.step_host(hb)
        n1 = s.step_launches()
        _, enq, wait = s.step_host_stats()
        print("   %5.2f s  steps %6d..  enqueue %4.1f us  wait %5.1f us  (launches per step %d)" % (time.perf_counter() - t0, b * 1000, enq, wait, n1), flush=True)
s.close()
