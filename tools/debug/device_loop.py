"""the device-resident RL loop (f110_episode_step_device + f110_episode_reset_done_device, nothing read in between) over batch
sizes: step_groups 1 (one block) against 0 (automatic)"""
import sys, os, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from _util import load_map_image, bench_start_poses
import f1tenth_gym_amd as amd
from f1tenth_gym_amd import build
img, res, origin = load_map_image("example_map")
print("# csrc", build.src_hash())


def run(E, A, G, n=400):
    s = amd.BatchSim(num_envs=E, num_agents=A, step_groups=G)
    s.set_map_image(img, res, origin); s.set_noise_rng(12345, 0.01); s.noise_prepare(800)
    poses = bench_start_poses(E, A)
    s.episode_init(0); s.episode_reset(poses)
    rng = np.random.default_rng(0)
    act = s.device_array((E * A, 2)); act.upload(np.stack([rng.uniform(-0.2, 0.2, E * A), rng.uniform(2, 6, E * A)], axis=1))
    for _ in range(100):
        s.episode_step_device(act); s.episode_reset_done_device()
    best = 1e9
    for rep in range(3):
        s.sync(); t0 = time.perf_counter()
        for _ in range(n):
            s.episode_step_device(act); s.episode_reset_done_device()
        s.sync(); best = min(best, (time.perf_counter() - t0) / n * 1e3)
    blocks = s.step_groups()[2]
    s.close()
    return best, blocks


for A, sizes in ((2, (2048, 4096, 8192, 16384, 32768, 65536)), (4, (4096, 16384, 65536)), (1, (16384, 65536))):
    for N in sizes:
        a, _ = run(N // A, A, 1); b, blk = run(N // A, A, 0)
        print("A %d agents %6d  one block %.4f ms  automatic %.4f ms (%d block%s)  %+.1f %%" % (A, N, a, b, blk, "s" if blk > 1 else "", (a / b - 1) * 100)); sys.stdout.flush()
