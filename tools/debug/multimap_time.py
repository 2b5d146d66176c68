"""step time of the per-env-map kernel with every env on slot 0 against the standard kernel"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import f1tenth_gym_amd as amd
from _util import load_map_image, bench_start_poses
img, res, origin = load_map_image("example_map")
E, A = 32768, 2
for mode in ("standard", "per-env maps (all slot 0)", "per-env maps (2 copies alternating)"):
    s = amd.BatchSim(num_envs=E, num_agents=A)
    s.set_map_image(img, res, origin)
    if mode != "standard":
        if "2 copies" in mode:
            s.add_map_image(img, res, origin)
            s.set_env_maps(np.arange(E) % 2)
        else:
            s.set_env_maps(np.zeros(E, dtype=np.int32))
    s.set_noise_table(np.random.default_rng(1).normal(0, 0.01, size=(200, 1080)))
    poses = bench_start_poses(E, A)
    s.reset(poses)
    rng = np.random.default_rng(0)
    d_act = s.device_array((E * A, 2)); d_act.upload(np.stack([rng.uniform(-0.2, 0.2, E * A), rng.uniform(2, 6, E * A)], axis=1))
    d_start = s.device_array((E * A, 3)); d_start.upload(poses)
    s.set_auto_reseat(d_start, 0, None)
    for _ in range(20): s.step_device(d_act)
    s.sync(); t0 = time.perf_counter()
    for _ in range(150): s.step_device(d_act)
    s.sync(); dt = (time.perf_counter() - t0) / 150
    print("%-40s %.3f ms/step  %.1f M agent-steps/s" % (mode, dt * 1e3, E * A / dt / 1e6))
    s.close()
