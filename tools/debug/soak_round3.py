"""long-run identity check of round 3's step (experimental build): the product's dispatch — pair test inside the
workgroup-flattened finalize, device-drawn noise with episodes that outlive a 256-row cache — against round 1's form
(k_collide on the side stream + k_finalize with fixed lanes, the one-wave k_integrate, no longest-first list, uploaded NumPy noise rows), same inputs, many steps,
every array compared at checkpoints.
    gpurun -- 'F110_LIB_VARIANT=experimental python tools/debug/soak_round3.py 32768 3000'"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import f1tenth_gym_amd as amd
from f1tenth_gym_amd import build
from _util import load_map_image, bench_start_poses
E = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
A = 2
img, res, origin = load_map_image("example_map")
poses = bench_start_poses(E, A)
sims = []
for which in ("product dispatch", "round-1 form"):
    s = amd.BatchSim(num_envs=E, num_agents=A, exp=({} if which == "product dispatch" else {"collide_mode": 0, "integrate_duo": 0, "task_order": 0}))
    s.set_map_image(img, res, origin)
    if which == "product dispatch":
        s.set_noise_rng(12345, 0.01, cache_rows=256)
    else:
        s.set_noise_table(np.random.default_rng(12345).normal(0, 0.01, size=(4096, 1080)))
    d_start = s.device_array((E * A, 3)); d_start.upload(poses)
    d_cnt = s.device_array((1,), dtype=np.int32); d_cnt.upload(np.zeros(1, dtype=np.int32))
    s.reset_device(d_start)
    s.set_auto_reseat(d_start, 0, d_cnt)
    sims.append((s, d_start, d_cnt, s.device_array((E * A, 2))))
print("csrc", build.src_hash(), "agents", E * A, "steps", T)
rng = np.random.default_rng(0)
t0 = time.time(); bad = 0
for t in range(T):
    if t % 20 == 0:
        act = np.stack([rng.uniform(-0.25, 0.25, E * A), rng.uniform(2.0, 7.0, E * A)], axis=1)
        for s, _, _, da in sims: da.upload(act)
    for s, _, _, da in sims: s.step_device(da)
    if t % 250 == 249 or t == T - 1:
        a, b = (s.get("state", "collisions", "collision_idx", "in_collision", "step_count", "scans") for s, _, _, _ in sims)
        for key in a:
            if not np.array_equal(a[key], b[key]):
                bad += 1; print("MISMATCH step", t, key, int(np.sum(a[key] != b[key])))
        print("step %d ok=%s resets %d / %d  (%.1f s)" % (t + 1, bad == 0, int(sims[0][2].download()[0]), int(sims[1][2].download()[0]), time.time() - t0))
print("DONE bad=%d rays compared per checkpoint %d" % (bad, E * A * 1080))
