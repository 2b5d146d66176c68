"""which march did the rays take?  (run on the GPU box)"""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import f1tenth_gym_amd as amd
from _util import load_map_image, raceline
img, res, origin = load_map_image("example_map")
w = raceline()
for layout in (3, 0):
    s = amd.BatchSim(num_envs=64, num_agents=2, map_layout=layout)
    s.set_map_image(img, res, origin)
    s.scan_path_stats(enable=True)
    k = (np.arange(128) * 37) % w.shape[0]
    poses = np.stack([w[k, 1], w[k, 2], w[k, 3] + np.pi / 2], axis=1)
    s.scan_batch(poses)
    print("layout", layout, "unit", s.scan_path_stats())
    s.reset(poses)
    for _ in range(5):
        s.step(np.tile([0.0, 3.0], (128, 1)))
    print("layout", layout, "step", s.scan_path_stats())
    s.close()
