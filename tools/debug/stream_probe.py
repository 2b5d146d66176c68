"""the bench-like step loop (300 pre-roll + 100 timed steps, in-step re-seats) under lab switches (experimental build): F110_LIB_VARIANT=experimental python tools/debug/stream_probe.py N [k=v ...]  (round 5: written for k_scan_stream_agent; rounds 5-6 timed the tiled / row-pair tables with it)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import f1tenth_gym_amd as amd
from _util import bench_start_poses, load_map_image
img, res, origin = load_map_image("example_map")
N = int(sys.argv[1]); exp = {k: int(v) for k, v in (a.split("=") for a in sys.argv[2:])}
E = N // 2
s = amd.BatchSim(num_envs=E, num_agents=2, exp=exp)
s.set_map_image(img, res, origin); s.set_noise_rng(12345, 0.01)
poses = bench_start_poses(E, 2)
s.reset(poses)
rng = np.random.default_rng(1)
d_act = [s.device_array((N, 2)) for _ in range(8)]
for d in d_act:
    d.upload(np.stack([rng.uniform(-0.2, 0.2, N), rng.uniform(2.0, 6.0, N)], axis=1))
d_start = s.device_array((N, 3)); d_start.upload(poses)
s.set_auto_reseat(d_start, 0)          # finished envs are re-seated inside the step, as in bench.py's timed region
for t in range(300):
    s.step_device(d_act[(t // 20) % 8])
s.sync()
K = 100
t0 = time.perf_counter()
for t in range(K):
    s.step_device(d_act[(t // 20) % 8])
s.sync()
dt = (time.perf_counter() - t0) / K
print("agents %6d %-60s %.4f ms/step  %.1f M agent-steps/s" % (N, exp, dt * 1e3, N / dt / 1e6), flush=True)
s.close()
