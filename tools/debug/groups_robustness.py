"""does the 2-group step keep its gain when the handle is not the first one of the process?"""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from _util import load_map_image, bench_start_poses
import f1tenth_gym_amd as amd
img, res, origin = load_map_image("example_map")


def run(N, G, keep=None):
    E, A = N // 2, 2
    s = amd.BatchSim(num_envs=E, num_agents=A, step_groups=G); s.set_map_image(img, res, origin); s.set_noise_rng(12345, 0.01); s.noise_prepare(400)
    poses = bench_start_poses(E, A); d = s.device_array((E * A, 3)); d.upload(poses); s.reset_device(d); s.set_auto_reseat(d, 0, None)
    rng = np.random.default_rng(0)
    act = s.device_array((E * A, 2)); act.upload(np.stack([rng.uniform(-0.2, 0.2, E * A), rng.uniform(2, 6, E * A)], axis=1))
    for t in range(30): s.step_device(act)
    s.sync(); t0 = time.perf_counter()
    for t in range(300): s.step_device(act)
    s.sync(); ms = (time.perf_counter() - t0) / 300 * 1e3
    if keep is None:
        s.close()
    else:
        keep.append(s)
    return ms


alive = []
for rep in range(3):
    for N in (65536, 16384, 4096):
        print("rep %d N %6d  G=1 %.4f ms   G=2 %.4f ms   G=2 (3 other handles alive) %s" % (
            rep, N, run(N, 1), run(N, 2), "%.4f ms" % run(N, 2) if len(alive) >= 3 else "-"))
    alive.append(None)
    run(1024, 1, keep=alive)   # leave a handle (2 streams) alive
    alive = [a for a in alive if a is not None]
