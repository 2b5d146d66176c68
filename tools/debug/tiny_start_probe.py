"""From the host's side: when does k_step_tiny's first workgroup start, when is the completion word seen?  (lab build; the first workgroup
stores the step's sequence number into a second word of page-locked memory as it starts, f110_step_host polls that word, then the
completion word.)    F110_LIB_VARIANT=experimental python tools/debug/tiny_start_probe.py [agents per env = 2] [steps = 5000]"""
import os, sys
os.environ.setdefault("F110_LIB_VARIANT", "experimental")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import f1tenth_gym_amd as amd
from f1tenth_gym_amd import workload

A = int(sys.argv[1]) if len(sys.argv) > 1 else 2
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
s = amd.BatchSim(num_envs=1, num_agents=A)
s.set_map(workload.map_stem("example_map") + ".yaml", ".png"); s.set_noise_rng(12345, 0.01)
s.reset(workload.bench_start_poses(1, A))
hb = s.host_block(("scans", "state", "agent_poses", "collisions", "collision_idx", "in_collision"))
hb.actions[...] = np.tile([0.05, 3.0], (A, 1))
for _ in range(STEPS + 300):      # (the episode's noise rows, once)
    s.step_host(hb)
s.reset(workload.bench_start_poses(1, A))
s.step_host_stats()
for _ in range(STEPS):
    s.step_host(hb)
_, enq, wait = s.step_host_stats()
print("1 env x %d car(s), plain loop: enqueue %.1f us, wait %.1f us" % (A, enq, wait))
MODES = {1: "the step", 2: "the same launch returning at once (start word, completion word, nothing else)"}
for mode in (1, 2, 1):
    s.reset(workload.bench_start_poses(1, A))
    s.exp_set("tiny_start_probe", mode)
    s.step_host_stats()
    for _ in range(STEPS):
        s.step_host(hb)
    _, enq, wait = s.step_host_stats()
    print("-- %-85s in the call: enqueue %.1f us + wait %.1f us" % (MODES[mode], enq, wait), flush=True)
    s.exp_set("tiny_start_probe", 0)
s.close()
