import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _util import load_map_image, oracle_map_dt
from oracle import orc
import f1tenth_gym_amd as amd
rng = np.random.default_rng(17)
mapname, integ, ld, layout = "berlin", 1, 0.0, 0
img, res, origin = load_map_image(mapname); dt, _, _ = oracle_map_dt(mapname)
E, A, T = 9, 3, 50
noise = np.random.default_rng(12345).normal(0., 0.01, size=(7, 1080))
p2 = dict(amd.DEFAULT_PARAMS); p2.update({'mu': 0.8, 'm': 3.2, 'length': 0.50, 'width': 0.28, 'a_max': 7.0})
variant = sys.argv[1] if len(sys.argv) > 1 else "full"
s = amd.BatchSim(num_envs=E, num_agents=A, integrator=integ, lidar_dist=ld, map_layout=layout)
s.set_map_image(img, res, origin)
ref = orc.SimOracle(E, A, integrator=integ, lidar_dist=ld); ref.set_map_dt(dt, res, origin)
if "nonoise" not in variant:
    s.set_noise_table(noise); ref.set_noise(noise)
if "noparams" not in variant:
    s.set_params(p2, 1); ref.set_params(p2, 1)
poses = np.stack([rng.uniform(-0.6, 0.6, E * A), rng.uniform(-0.6, 0.6, E * A), rng.uniform(0, 2 * np.pi, E * A)], axis=1)
s.reset(poses); ref.reset(poses)
for t in range(T):
    if t % 10 == 0:
        act = np.stack([rng.uniform(-0.4, 0.4, E * A), rng.uniform(-2.0, 5.0, E * A)], axis=1)
    s.step(act); ref.step(act, 8)
    o = s.get("scans", "state", "in_collision", "agent_poses")
    d = np.abs(o["scans"] - ref.scans)
    if d.max() > 1e-9:
        i, b = np.unravel_index(np.argmax(d), d.shape)
        bad = np.nonzero(d[i] > 1e-9)[0]
        print("step", t, "agent", i, "env", i // A, "slot", i % A, "beam", b, "gpu", o["scans"][i, b], "ref", ref.scans[i, b], "nbad", len(bad), "bad beams", bad[:10], bad[-3:])
        print(" wall flags env:", o["in_collision"][(i // A) * A:(i // A + 1) * A], "state", o["state"][i])
        print(" poses env:", o["agent_poses"][(i // A) * A:(i // A + 1) * A])
        # recompute with oracle pieces
        ego = np.array([o["state"][i, 0], o["state"][i, 1], o["state"][i, 4]])
        par = p2 if (i % A == 1 and "noparams" not in variant) else amd.DEFAULT_PARAMS
        for jj in range(A):
            if jj == i % A: continue
            v = orc.get_vertices(o["agent_poses"][(i // A) * A + jj], par['length'], par['width'])
            print("  opp", jj, "ref window", orc.get_blocked_view_indices(ego, v, s.scan_angles))
        break
else:
    print("no mismatch")
