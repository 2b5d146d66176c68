"""Worker of tests/test_gpu_round6.py::test_sharded_vec_env_gathers_the_observation_on_one_device: ShardedVecEnv(devices=[0, 0, 0, 0],
gather_obs=True) — four handles on device 0, tests/rccl_stub/librccl.so.1 standing in for RCCL — against ONE F110VecEnv handle
stepping the same envs: after every step each shard's receive buffers hold EVERY shard's scans and scalars, i.e. the single
handle's observation cut into blocks.  All-gather f64, then float32 to root 2.  Prints RESULT {...}."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import f1tenth_gym_amd as amd  # noqa: E402
from f1tenth_gym_amd import workload  # noqa: E402

K, E, A, T = 4, 48, 2, 12
kw = dict(map=workload.map_stem("example_map"), map_ext=".png", num_agents=A, obs_fields=("scans", "poses_x", "poses_y", "poses_theta",
                                                                                                        "linear_vels_x", "ang_vels_z", "collisions"))
poses = workload.bench_start_poses(E, A, gap_wp=4).reshape(E, A, 3)
errors, checks = [], 0
try:
    for f32, root in ((False, None), (True, 2)):
        one = amd.F110VecEnv(E, device_logic=True, **kw)
        sh = amd.ShardedVecEnv(E, devices=[0] * K, gather_obs=True, gather_f32=f32, gather_root=root, **kw)
        o1 = one.reset(poses); o2 = sh.reset(poses)
        rng = np.random.default_rng(3)
        for t in range(T):
            act = np.stack([rng.uniform(-0.3, 0.3, (E, A)), rng.uniform(1.0, 7.0, (E, A))], axis=2)
            o1 = one.step(act); o2 = sh.step(act)
            for key in o1[0]:
                assert np.array_equal(np.asarray(o1[0][key]), np.asarray(o2[0][key])), (t, key)
            assert np.array_equal(o1[2], o2[2]), (t, "done")
            sh.sync()
            n = E // K * A
            want_s = o1[0]["scans"].reshape(K, n, -1)
            want_c = np.stack([o1[0][k].reshape(K, n) for k in ("poses_x", "poses_y", "poses_theta", "linear_vels_x")]
                              + [np.zeros((K, n))] + [o1[0][k].reshape(K, n) for k in ("ang_vels_z", "collisions")], axis=1)
            for k, (d_s, d_c) in enumerate(sh.gathered_views()):
                if root is not None and k != root:
                    continue
                got_s, got_c = d_s.download(), d_c.download()
                assert np.array_equal(got_s, want_s.astype(np.float32) if f32 else want_s), (t, k, "gathered scans")
                assert np.array_equal(got_c, want_c), (t, k, "gathered scalars")
                checks += 1
        sh.close(); one.sim.batch.close()
except BaseException as ex:  # noqa: BLE001
    import traceback
    errors.append("%r\n%s" % (ex, traceback.format_exc()[-1500:]))
print("RESULT " + json.dumps({"errors": errors, "checks": checks}))
sys.stdout.flush()
os._exit(0 if not errors else 1)
