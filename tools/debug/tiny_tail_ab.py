"""(lab build) k_step_tiny's tail for one env of two cars: finalize_duo_tiny (the product) against finalize_pair_body (the general body the
kernel used before), alternating in ONE process on ONE handle — BatchSim.step_host per step, rows of the noise cache warm.
    F110_LIB_VARIANT=experimental python tools/debug/tiny_tail_ab.py [rounds=4]"""
import os, sys, time
os.environ.setdefault("F110_LIB_VARIANT", "experimental")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import f1tenth_gym_amd as amd
from f1tenth_gym_amd import workload

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
s = amd.BatchSim(num_envs=1, num_agents=2)
s.set_map(workload.map_stem("example_map") + ".yaml", ".png"); s.set_noise_rng(12345, 0.01)
poses = workload.bench_start_poses(1, 2)
hb = s.host_block(("scans", "state", "agent_poses", "collisions", "collision_idx", "in_collision"))
hb.actions[...] = np.array([[0.05, 3.0], [-0.05, 2.5]])
s.reset(poses)
for _ in range(4300):
    s.step_host(hb)
res = {0: [], 1: []}
for rnd in range(R):
    for general in (0, 1):
        s.exp_set("tiny_general_tail", general)
        s.reset(poses)
        for _ in range(200):
            s.step_host(hb)
        s.step_host_stats()
        for _ in range(4000):
            s.step_host(hb)
        _, enq, wait = s.step_host_stats()
        res[general].append(enq + wait)
        print("round %d  %-22s in the call: enqueue %4.1f + wait %5.1f us" % (rnd, "finalize_pair_body" if general else "finalize_duo_tiny", enq, wait), flush=True)
print("in the call, mean over the rounds: finalize_duo_tiny %.2f us, finalize_pair_body %.2f us" % (np.mean(res[0]), np.mean(res[1])))
s.close()
