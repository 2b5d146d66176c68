"""Where k_step_tiny's time goes: the kernel's phase stamps (100 MHz clock; lab build, exp tiny_trace_hi / _lo) over a loop of
F110Env-shaped host steps (1 env x A cars), next to what the host sees (enqueue, wait).
    F110_LIB_VARIANT=experimental python tools/debug/tiny_timeline.py [agents per env = 2] [steps = 2000]"""
import os, sys, time
os.environ.setdefault("F110_LIB_VARIANT", "experimental")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import f1tenth_gym_amd as amd
from f1tenth_gym_amd import workload

A = int(sys.argv[1]) if len(sys.argv) > 1 else 2
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
s = amd.BatchSim(num_envs=1, num_agents=A)
s.set_map(workload.map_stem("example_map") + ".yaml", ".png"); s.set_noise_rng(12345, 0.01)
s.reset(workload.bench_start_poses(1, A))
hb = s.host_block(("scans", "state", "agent_poses", "collisions", "collision_idx", "in_collision"))
hb.actions[...] = np.tile([0.05, 3.0], (A, 1))
WG = (A * 17 + 3) // 4
tr = s.device_array((WG, 16), dtype=np.uint64)
tr.upload(np.zeros((WG, 16), dtype=np.uint64))
s.exp_set("tiny_trace_hi", int(np.array(tr.ptr >> 32, dtype=np.uint32).view(np.int32)))
s.exp_set("tiny_trace_lo", int(np.array(tr.ptr & 0xffffffff, dtype=np.uint32).view(np.int32)))
for _ in range(200):
    s.step_host(hb)
rows = []
s.step_host_stats()
t0 = time.perf_counter()
for _ in range(STEPS):
    s.step_host(hb)
    assert s.step_launches() == 1
    r = tr.download().astype(np.int64)        # (a d2h copy per step: the loop is slower than the real one; the stamps are the kernel's own)
    first = r[:, 0].min()
    last = int(np.argmax(r[:, 4] > 0)) if (r[:, 4] > 0).any() else 0
    rows.append([(r[:, 0].max() - first), (r[:, 1] - first).max(), (r[:, 2] - first).max(), (r[:, 3] - first).max(),
                 r[last, 4] - first, r[last, 5] - first, r[last, 6] - first, r[last, 7] - first])
    tr.upload(np.zeros((WG, 16), dtype=np.uint64))
_, enq, wait = s.step_host_stats()
m = np.array(rows, dtype=np.float64) * 0.01   # 10 ns ticks -> us
names = ["last workgroup's first wave starts", "integration + header done (slowest workgroup)", "beams marched + stored (slowest)", "released (slowest)",
         "the last workgroup knows it is last", "(tail begins)", "finalize + host block issued", "completion word stored"]
print("k_step_tiny, 1 env x %d car(s), %d steps; microseconds after the FIRST workgroup's first wave started (mean / p10 / p90):" % (A, STEPS))
for i, nme in enumerate(names):
    print("  %-52s %6.1f  %6.1f  %6.1f" % (nme, m[:, i].mean(), np.percentile(m[:, i], 10), np.percentile(m[:, i], 90)))
print("host (this loop, with the trace copies in it): enqueue %.1f us, wait %.1f us per step" % (enq, wait))
s.close()
