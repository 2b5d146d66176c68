"""which march the ray pass's rays take (diagnostics): F110_LIB_VARIANT=experimental python tools/debug/ray_pass_stats.py [thr]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import f1tenth_gym_amd as amd
from _util import load_map_image, bench_start_poses
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 96
E, A = 2048, 2
for rp in (0, 1):
    s = amd.BatchSim(num_envs=E, num_agents=A, exp={"ray_pass": rp, "ray_thr": thr})
    s.set_map_image(*load_map_image("example_map"))
    s.set_noise_rng(12345, 0.01); s.noise_prepare(700)
    poses = bench_start_poses(E, A)
    rng = np.random.default_rng(0)
    d_act = s.device_array((E * A, 2)); d_act.upload(np.stack([rng.uniform(-0.2, 0.2, E * A), rng.uniform(2, 6, E * A)], axis=1))
    d_start = s.device_array((E * A, 3)); d_start.upload(poses)
    s.reset_device(d_start); s.set_auto_reseat(d_start, 0, None)
    for _ in range(300):
        s.step_device(d_act)
    s.sync()
    s.scan_path_stats(enable=True, read=True)
    t0 = time.perf_counter()
    for _ in range(200):
        s.step_device(d_act)
    s.sync(); dt = (time.perf_counter() - t0) / 200
    st = s.scan_path_stats(enable=False)
    rays = st["fast"] / 200.0 - E * A * 1080   # (k_integrate adds num_beams per fast scan to the same counter)
    print("ray_pass", rp, "thr", thr, "ms/step %.4f" % (dt * 1e3), "ray pass per step: rays %.1f, block fetches %.1f, samples %.1f" % (rays, st["guard"] / 200.0, st["exact"] / 200.0),
          "-> samples per ray %.1f, per block %.2f" % (st["exact"] / max(rays * 200, 1), st["exact"] / max(st["guard"], 1)))
    s.close()
