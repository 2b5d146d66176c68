"""two env groups (main + side stream) with the in-kernel pair tests: gain per A, and whether it survives handles created
earlier in the process / alive next to it.  F110_LIB_VARIANT=experimental python tools/debug/groups_multi.py"""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from _util import load_map_image, bench_start_poses
import f1tenth_gym_amd as amd
from f1tenth_gym_amd import build
img, res, origin = load_map_image("example_map")
print("# csrc", build.src_hash())


def run(N, A, G, keep=None, split=0, steps=200):
    E = N // A
    s = amd.BatchSim(num_envs=E, num_agents=A, step_groups=G, exp={"group_split": split} if split else None)
    s.set_map_image(img, res, origin); s.set_noise_rng(12345, 0.01); s.noise_prepare(400)
    poses = bench_start_poses(E, A); d = s.device_array((E * A, 3)); d.upload(poses); s.reset_device(d); s.set_auto_reseat(d, 0, None)
    rng = np.random.default_rng(0)
    act = s.device_array((E * A, 2)); act.upload(np.stack([rng.uniform(-0.2, 0.2, E * A), rng.uniform(2, 6, E * A)], axis=1))
    for t in range(60): s.step_device(act)
    s.sync(); t0 = time.perf_counter()
    for t in range(steps): s.step_device(act)
    s.sync(); ms = (time.perf_counter() - t0) / steps * 1e3
    if G > 1: PROBES.append(s.step_groups())
    if keep is None:
        s.close()
    else:
        keep.append(s)
    return ms


alive = []
PROBES = []
for rep in range(4):
    for A in (16, 8, 4, 3, 2, 1):
        N = 65536 if A != 3 else 65535
        del PROBES[:]
        print("rep %d A %2d N %6d  G=1 %.4f   G=2 %.4f   G=2 60/40 %.4f   G=2 70/30 %.4f  ms per step   (%d other handles alive; (groups, probes) %s)" % (
            rep, A, N, run(N, A, 1), run(N, A, 2), run(N, A, 2, split=60), run(N, A, 2, split=70), len(alive), PROBES))
    for A, N in ((2, 4096), (2, 16384), (4, 4096)):
        del PROBES[:]
        print("rep %d A %2d N %6d  G=1 %.4f   G=2 %.4f  %s" % (rep, A, N, run(N, A, 1, steps=600), run(N, A, 2, steps=600), PROBES))
    run(1024, 2, 1, keep=alive)   # leave a handle alive
    sys.stdout.flush()
