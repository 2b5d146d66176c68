"""k_scan_stream_agent (lane refill, round 5) against k_scan_rays_agent: bit-identical outputs, then timings.
Run on the GPU box with the experimental build:
    F110_LIB_VARIANT=experimental python tools/debug/stream_ab.py [parity] [time]
parity: same seeds through scan_stream = 0 / 1 (several refill thresholds, batch sizes, agents per env, a yawed map, per-env maps,
the lookup counter) — every output compared exactly.   time: steps per second at 65 536 / 32 768 / 16 384 agents, both kernels."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import f1tenth_gym_amd as amd  # noqa: E402
from _util import bench_start_poses, load_map_image  # noqa: E402

ALL = ("scans", "state", "collisions", "collision_idx", "in_collision")
img, res, origin = load_map_image("example_map")


def run(E, A, T, exp, origin_=None, beams=1080, per_env=False, lookups=False, seed=0):
    s = amd.BatchSim(num_envs=E, num_agents=A, num_beams=beams, exp=exp)
    s.set_map_image(img, res, origin_ or origin)
    if per_env:
        b = load_map_image("berlin")
        slot = s.add_map_image(*b)
        s.set_env_maps(np.arange(E) % 2 * slot)
    s.set_noise_rng(12345, 0.01)
    poses = bench_start_poses(E, A)
    if origin_ is not None:      # the same cars on the yawed map: rotate the start poses with it
        c, sn = np.cos(origin_[2]), np.sin(origin_[2])
        dx, dy = poses[:, 0] - origin[0], poses[:, 1] - origin[1]
        poses = np.stack([origin_[0] + c * dx - sn * dy, origin_[1] + sn * dx + c * dy, poses[:, 2] + origin_[2]], axis=1)
    if per_env:
        poses = poses.reshape(E, A, 3).copy()
        poses[1::2] = [[0.0, 0.0, 0.3], [0.9, 0.5, 2.0]][:A] if A <= 2 else poses[1::2]
        poses = poses.reshape(-1, 3)
    s.reset(poses)
    if lookups:
        s.scan_lookup_count(enable=True, read=False)
    rng = np.random.default_rng(seed)
    outs = []
    for t in range(T):
        act = np.stack([rng.uniform(-0.3, 0.3, E * A), rng.uniform(1.0, 7.0, E * A)], axis=1)
        s.step(act)
        if t % 7 == 0 or t == T - 1:
            outs.append(s.get(*ALL))
    lk = s.scan_lookup_count() if lookups else None
    s.close()
    return outs, lk


def same(a, b):
    return all(np.array_equal(x[k], y[k]) for x, y in zip(a, b) for k in ALL)


if "parity" in sys.argv or len(sys.argv) == 1:
    cases = [dict(E=32, A=2, T=40), dict(E=700, A=1, T=25), dict(E=4096, A=2, T=25), dict(E=33, A=3, T=30), dict(E=1500, A=4, T=20),
             dict(E=48, A=2, T=30, origin_=[-3.0, -4.0, 0.3]), dict(E=64, A=2, T=30, per_env=True), dict(E=256, A=2, T=15, lookups=True),
             dict(E=40, A=2, T=20, beams=271), dict(E=16384, A=2, T=8)]
    for c in cases:
        ref, lk0 = run(exp={"scan_stream": 0}, **c)
        for R in (32, 16, 1, 64):
            got, lk1 = run(exp={"scan_stream": 1, "stream_refill": R}, **c)
            ok = same(ref, got) and lk0 == lk1
            print("%-70s refill %2d: %s%s" % (c, R, "identical" if ok else "DIFFERENT", "" if lk0 is None else "  lookups %d / %d" % (lk0, lk1)), flush=True)
            assert ok

if "time" in sys.argv:
    for N in (65536, 32768, 16384):
        for exp in ({"scan_stream": 0}, {"scan_stream": 1, "stream_refill": 32}, {"scan_stream": 1, "stream_refill": 16}, {"scan_stream": 1, "stream_refill": 48}):
            E = N // 2
            s = amd.BatchSim(num_envs=E, num_agents=2, exp=exp)
            s.set_map_image(img, res, origin); s.set_noise_rng(12345, 0.01)
            s.reset(bench_start_poses(E, 2))
            rng = np.random.default_rng(1)
            d_act = [s.device_array((N, 2)) for _ in range(8)]
            for d in d_act:
                d.upload(np.stack([rng.uniform(-0.2, 0.2, N), rng.uniform(2.0, 6.0, N)], axis=1))
            for t in range(300):
                s.step_device(d_act[(t // 20) % 8])
            s.sync()
            t0 = time.perf_counter()
            K = 200
            for t in range(K):
                s.step_device(d_act[(t // 20) % 8])
            s.sync()
            dt = (time.perf_counter() - t0) / K
            print("agents %6d %-45s %.4f ms/step  %.1f M agent-steps/s" % (N, exp, dt * 1e3, N / dt / 1e6), flush=True)
            s.close()
