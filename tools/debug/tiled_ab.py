"""the step's scan on a 4x4-tiled copy (exp pad_tiled=1) and on a row-pair copy (pad_tiled=2, round 6) of the PADDED table (experimental build) against the row-major one: bit-identical outputs.
    F110_LIB_VARIANT=experimental python tools/debug/tiled_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import f1tenth_gym_amd as amd
from _util import bench_start_poses, load_map_image
ALL = ("scans", "state", "collisions", "collision_idx", "in_collision")
for mapname, E, A, T in (("example_map", 64, 2, 40), ("berlin", 33, 3, 30), ("example_map", 1024, 2, 12), ("skirk", 40, 1, 30), ("example_map", 16384, 2, 6)):
    outs = []
    for pt in (0, 1, 2):     # row-major, 4x4 tiles, row pairs (round 6)
        s = amd.BatchSim(num_envs=E, num_agents=A, exp={"pad_tiled": pt})
        s.set_map_image(*load_map_image(mapname)); s.set_noise_rng(12345, 0.01)
        rng = np.random.default_rng(2)
        if mapname == "example_map":
            poses = bench_start_poses(E, A)
        else:
            poses = np.stack([rng.uniform(-0.6, 0.6, E * A), rng.uniform(-0.6, 0.6, E * A), rng.uniform(0, 6.28, E * A)], axis=1)
        s.reset(poses)
        d = s.device_array((E * A, 2))
        rec = []
        for t in range(T):
            d.upload(np.stack([rng.uniform(-0.3, 0.3, E * A), rng.uniform(1.0, 7.0, E * A)], axis=1))
            s.step_device(d)
            if t % 5 == 0 or t == T - 1:
                rec.append(s.get(*ALL))
        outs.append(rec); s.close()
    ok = all(np.array_equal(a[k], b[k]) for other in outs[1:] for a, b in zip(outs[0], other) for k in ALL)
    print("%-12s E %5d A %d: %s" % (mapname, E, A, "identical" if ok else "DIFFERENT"), flush=True)
    assert ok
