"""two env groups against one block over the batch size (A = 1, 2, 4), and on the host-synchronised drop-in path.
F110_LIB_VARIANT=experimental python tools/debug/groups_sizes.py"""
import sys, os, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from _util import load_map_image, bench_start_poses, MAPS
import f1tenth_gym_amd as amd
from f1tenth_gym_amd import build
img, res, origin = load_map_image("example_map")
print("# csrc", build.src_hash())


def run(N, A, G, steps=300):
    E = N // A
    s = amd.BatchSim(num_envs=E, num_agents=A, step_groups=G)
    s.set_map_image(img, res, origin); s.set_noise_rng(12345, 0.01); s.noise_prepare(600)
    poses = bench_start_poses(E, A); d = s.device_array((E * A, 3)); d.upload(poses); s.reset_device(d); s.set_auto_reseat(d, 0, None)
    rng = np.random.default_rng(0)
    act = s.device_array((E * A, 2)); act.upload(np.stack([rng.uniform(-0.2, 0.2, E * A), rng.uniform(2, 6, E * A)], axis=1))
    for t in range(100): s.step_device(act)
    best = 1e9
    for rep in range(3):
        s.sync(); t0 = time.perf_counter()
        for t in range(steps): s.step_device(act)
        s.sync(); best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    g = s.step_groups()
    s.close()
    return best, g


for A in (2, 1, 4):
    for N in (256, 512, 1024, 2048, 4096, 8192, 16384, 24576, 32768, 49152, 65536, 131072, 262144):
        a, _ = run(N, A, 1); b, g = run(N, A, 2)
        print("A %d N %6d  one block %.4f ms   two groups %.4f ms   %+.1f %%   %s" % (A, N, a, b, (a / b - 1) * 100, g)); sys.stdout.flush()


def vec(E, G, n=1500):
    env = amd.F110VecEnv(E, auto_reset=True, device_logic=True, obs_fields=(), step_groups=G, map=os.path.join(MAPS, "example_map"), map_ext=".png")
    env.reset(bench_start_poses(E, 2).reshape(E, 2, 3))
    rng = np.random.default_rng(0)
    env.action_buffer[...] = np.stack([rng.uniform(-0.2, 0.2, (E, 2)), rng.uniform(2, 6, (E, 2))], axis=2)
    for _ in range(100): env.step(None)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n): env.step(None)
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    g = env.sim.batch.step_groups() if hasattr(env, "sim") else None
    env.close() if hasattr(env, "close") else None
    return best, g


for E in (256, 1024, 2048, 8192, 32768):
    a, _ = vec(E, 1, 1500 if E <= 8192 else 400); b, g = vec(E, 2, 1500 if E <= 8192 else 400)
    print("F110VecEnv E %5d  one block %.4f ms   two groups %.4f ms   %+.1f %%  %s" % (E, a, b, (a / b - 1) * 100, g)); sys.stdout.flush()
