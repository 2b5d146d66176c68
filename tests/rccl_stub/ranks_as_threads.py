"""Worker of tests/test_gpu_round4.py::test_observation_gather_with_several_ranks_on_one_device: two THREADS are the two ranks
(tests/rccl_stub/librccl.so.1 stands in for RCCL, see its header), each with its own BatchSim handle on device 0 and its own
shard of envs.  Every gather form is checked on both ranks against what the peer really holds at that step:
all-gather f64 in the step's stream, overlapped with the next step (double-buffered), float32 transport, gather-to-root
(root 0 and root 1), and root + float32 + overlap.  Prints RESULT {...}."""
import json, os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import f1tenth_gym_amd as amd
from _util import load_map_image, bench_start_poses

WORLD = int(sys.argv[1]) if len(sys.argv) > 1 else 2
E, A, B, T = 24 if WORLD <= 2 else 6, 2, 1080, 6 if WORLD <= 2 else 4
N = E * A
img, res, origin = load_map_image("example_map")
uid = amd.BatchSim.comm_unique_id()
barrier = threading.Barrier(WORLD)
truth = [dict() for _ in range(WORLD)]     # truth[rank][(leg, t)] = (scans, scalars) that rank held after step t of that leg
errors, checks = [], [0] * WORLD


def scalars_of(o):
    return np.stack([o["poses_x"], o["poses_y"], o["poses_theta"], o["linear_vels_x"], np.zeros(N), o["ang_vels_z"], o["collisions"]])


def rank_main(rank):
    try:
        s = amd.BatchSim(num_envs=E, num_agents=A)
        s.set_map_image(img, res, origin); s.set_noise_rng(12345 + rank, 0.01)
        poses = bench_start_poses(E * WORLD, A).reshape(WORLD, N, 3)[rank]       # this rank's shard of the envs
        s.reset(poses)
        s.comm_init(WORLD, rank, uid)
        assert s.comm_info() == (WORLD, rank)
        rng = np.random.default_rng(100 + rank)
        legs = (("gather", False, False, None), ("gather_overlap", True, False, None), ("gather_f32", False, True, None),
                ("gather_f32_overlap", True, True, None), ("gather_root0", False, False, 0), ("gather_root1", False, False, WORLD - 1),
                ("gather_root1_f32_overlap", True, True, WORLD - 1))
        for leg, overlap, f32, root in legs:
            recv_here = root is None or root == rank
            s.comm_set_overlap(overlap)
            nbuf = 2 if overlap else 1
            rs = [s.device_array((WORLD if recv_here else 1, N, B), np.float32 if f32 else np.float64) for _ in range(nbuf)]
            rc = [s.device_array((WORLD if recv_here else 1, 7, N)) for _ in range(nbuf)]
            for d in rs + rc:
                d.upload(np.full(d.shape, -7.0, dtype=d.dtype))
            pending = None
            for t in range(T):
                s.step(np.stack([rng.uniform(-0.3, 0.3, N), rng.uniform(1, 7, N)], axis=1))
                if f32 or root is not None:
                    s.comm_gather_obs(rs[t % nbuf], rc[t % nbuf], f32=f32, root=root)
                else:
                    s.comm_all_gather_obs(rs[t % nbuf], rc[t % nbuf])
                o = s.get("scans", "poses_x", "poses_y", "poses_theta", "linear_vels_x", "ang_vels_z", "collisions")
                truth[rank][(leg, t)] = (o["scans"].astype(np.float32) if f32 else o["scans"], scalars_of(o))
                barrier.wait()            # the peer has published its truth for step t
                # overlapped: consume the PREVIOUS gather while this step's is in flight; else this step's
                todo = [] if overlap else [t]
                if overlap and pending is not None:
                    todo = [pending]
                pending = t
                if t == T - 1 and overlap:
                    todo.append(t)
                for tt in todo:
                    if recv_here:
                        got_s, got_c = rs[tt % nbuf].download(), rc[tt % nbuf].download()
                        for p in range(WORLD):
                            want_s, want_c = truth[p][(leg, tt)]
                            assert np.array_equal(got_s[p], want_s), (leg, tt, rank, p, "scans")
                            assert np.array_equal(got_c[p], want_c), (leg, tt, rank, p, "scalars")
                            checks[rank] += 1
                    else:
                        assert np.all(rs[tt % nbuf].download() == -7.0), (leg, tt, rank, "a non-root rank's buffer was written")
                barrier.wait()            # nobody steps on (and overwrites a buffer) before both have checked
            s.comm_set_overlap(False)
            for d in rs + rc:
                d.free()
        s.close()
    except BaseException as ex:  # noqa: BLE001
        import traceback
        errors.append("rank %d: %s\n%s" % (rank, ex, traceback.format_exc()[-1500:]))
        try:
            barrier.abort()
        except Exception:  # noqa: BLE001
            pass


threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(WORLD)]
for th in threads:
    th.start()
for th in threads:
    th.join(240)
alive = [th.is_alive() for th in threads]
print("RESULT " + json.dumps({"errors": errors, "checks": checks, "hung": alive, "world": WORLD, "steps": T}))
sys.stdout.flush()
os._exit(0 if not errors and not any(alive) else 1)
