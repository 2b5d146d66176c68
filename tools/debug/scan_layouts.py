"""kernel time of ScanSimulator2D.scan batches per map layout (run under rocprofv3 --kernel-trace --stats)"""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import f1tenth_gym_amd as amd
from _util import load_map_image
mapname = sys.argv[1] if len(sys.argv) > 1 else "berlin"
layout = int(sys.argv[2]) if len(sys.argv) > 2 else 3
img, res, origin = load_map_image(mapname)
s = amd.BatchSim(num_envs=1, num_agents=1, map_layout=layout)
s.set_map_image(img, res, origin)
dt = s.get_map_dt()
free = np.argwhere(dt > 0.4)
rng = np.random.default_rng(0)
sel = free[rng.choice(len(free), 16384)]
poses = np.stack([origin[0] + (sel[:, 1] + 0.5) * res, origin[1] + (sel[:, 0] + 0.5) * res, rng.uniform(-3, 3, len(sel))], axis=1)
for _ in range(5):
    s.scan_batch(poses)
print("done", mapname, layout)
