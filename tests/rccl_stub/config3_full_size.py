"""Worker of tests/test_gpu_round6.py::test_config3_full_size_eight_ranks_on_one_device — BASELINE configs[3] at its real size
(262 144 agents = 8 ranks x 16 384 envs x 2 agents, 1080 beams) on ONE device: the 8 ranks are threads (tests/rccl_stub/librccl.so.1
stands in for RCCL; RCCL itself refuses several ranks on one device), each with its own BatchSim handle on device 0 and its contiguous
block of the GLOBAL env axis (the reference's serial agent loop base_classes.py:568-585 is what is being sharded).

What is checked, every step, on every rank, at full size (8 x [8][32768][1080] float64 receive buffers, 2.26 GB each, double-buffered
in the overlapped leg):
  * every peer's block in MY receive buffers equals what that peer holds (16-byte digests over the 283 MB blocks, and the scalars);
  * the TWIN property ACROSS ranks through the gathered buffer: global envs g and g + 783 start on the same raceline waypoint
    ((g * 7919) mod 783) and get the same actions, so their rows are identical — and they live in different ranks' blocks for most
    g, which checks the block offsets independently of the digests;
  * the first 32 envs of every rank against the CPU oracle (flags exact, floats <= 1e-9).
Legs: all-gather f64 in the step's stream; overlapped (double-buffered); float32 to root 0 overlapped.  Prints RESULT {...}."""
import json
import os
import struct
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import f1tenth_gym_amd as amd  # noqa: E402
from f1tenth_gym_amd import workload  # noqa: E402
from oracle import orc  # noqa: E402
from _util import rel_err  # noqa: E402

WORLD = int(sys.argv[1]) if len(sys.argv) > 1 else 8
E = int(sys.argv[2]) if len(sys.argv) > 2 else 16384      # envs per rank
A, B, T, NREF = 2, 1080, 3, 32
N = E * A
img, res, origin = workload.load_map_image("example_map")
uid = amd.BatchSim.comm_unique_id()
barrier = threading.Barrier(WORLD)
digests = [dict() for _ in range(WORLD)]      # digests[rank][(leg, t)] = (scan digest f64, scan digest as f32, scalar digest)
errors, checks, times = [], [0] * WORLD, {}
mem_used = [0.0]


def digest(arr):
    w = np.ascontiguousarray(arr).view(np.uint64 if arr.dtype.itemsize == 8 else np.uint32).reshape(-1)
    return struct.pack("<QQ", int(np.bitwise_xor.reduce(w)), int(np.sum(w, dtype=np.uint64)))


def actions_for(genv, t):
    """actions of global envs `genv` at step t: a function of (genv mod 783, t) only, so twins get the same"""
    rng = np.random.default_rng(1000 + t)
    a783 = np.stack([rng.uniform(-0.2, 0.2, (783, A)), rng.uniform(2.0, 6.0, (783, A))], axis=2)
    return a783[genv % 783].reshape(len(genv) * A, 2)


def rank_main(rank):
    try:
        genv = workload.shard_envs(E, rank)
        s = amd.BatchSim(num_envs=E, num_agents=A, num_beams=B)
        s.set_map_image(img, res, origin); s.set_noise_rng(12345, 0.01)
        poses = workload.start_poses(genv, A)          # (g * 7919) mod 783: g and g + 783 are twins
        ref = orc.SimOracle(NREF, A, num_beams=B)
        ref.set_map_dt(s.get_map_dt(), res, origin)
        s.comm_init(WORLD, rank, uid)
        assert s.comm_info() == (WORLD, rank)
        for leg, overlap, f32, root in (("gather", False, False, None), ("gather_overlap", True, False, None), ("gather_root0_f32_overlap", True, True, 0)):
            recv_here = root is None or root == rank
            s.comm_set_overlap(overlap)
            nbuf = 2 if overlap else 1
            rs = [s.device_array((WORLD if recv_here else 1, N, B), np.float32 if f32 else np.float64) for _ in range(nbuf)]
            rc = [s.device_array((WORLD if recv_here else 1, 7, N)) for _ in range(nbuf)]
            s.reset(poses)
            ref.set_noise(np.random.default_rng(12345).normal(0., 0.01, size=(T + 1, B))); ref.reset(poses[:NREF * A])
            barrier.wait()
            if rank == 0:
                free_b, total_b = s.device_mem_info()
                mem_used[0] = max(mem_used[0], (total_b - free_b) / 1e9)
            t_leg = time.perf_counter()
            pending = None
            for t in range(T):
                act = actions_for(genv, t)
                s.step(act); ref.step(act[:NREF * A], 4)
                if f32 or root is not None:
                    s.comm_gather_obs(rs[t % nbuf], rc[t % nbuf], f32=f32, root=root)
                else:
                    s.comm_all_gather_obs(rs[t % nbuf], rc[t % nbuf])
                o = s.get("scans", "poses_x", "poses_y", "poses_theta", "linear_vels_x", "ang_vels_z", "collisions", "state", "in_collision")
                scal = np.stack([o["poses_x"], o["poses_y"], o["poses_theta"], o["linear_vels_x"], np.zeros(N), o["ang_vels_z"], o["collisions"]])
                digests[rank][(leg, t)] = (digest(o["scans"]), digest(o["scans"].astype(np.float32)), digest(scal))
                # this rank's first envs against the oracle
                n = NREF * A
                assert np.array_equal(o["collisions"][:n], ref.collisions) and np.array_equal(o["in_collision"][:n], ref.in_collision), (leg, t, rank, "flags")
                assert rel_err(o["state"][:n], ref.state) < 1e-9 and rel_err(o["scans"][:n], ref.scans) < 1e-9, (leg, t, rank, "oracle")
                barrier.wait()            # every rank has published its digests for step t
                todo = [] if overlap else [t]
                if overlap and pending is not None:
                    todo = [pending]
                pending = t
                if t == T - 1 and overlap:
                    todo.append(t)
                for tt in todo:
                    if not recv_here:
                        continue
                    buf_s, buf_c = rs[tt % nbuf], rc[tt % nbuf]
                    for p in range(WORLD):    # one 283 MB block at a time
                        blk = buf_s.download_part(p, 1)
                        want = digests[p][(leg, tt)]
                        assert digest(blk) == (want[1] if f32 else want[0]), (leg, tt, rank, p, "scans digest")
                        assert digest(buf_c.download_part(p, 1)) == want[2], (leg, tt, rank, p, "scalars digest")
                        checks[rank] += 1
                        if p == (rank + 1) % WORLD:
                            # twins across blocks: global env g (peer p's local row) and g + 783 (which may sit in the NEXT rank's block)
                            nxt = (p + 1) % WORLD
                            blk2 = buf_s.download_part(nxt, 1) if nxt != p else blk
                            g = workload.shard_envs(E, p)
                            tw = g + 783
                            for dst_rank, dst_blk in ((p, blk), (nxt, blk2)):
                                sel = (tw // E == dst_rank) & (tw < WORLD * E)
                                if sel.any():
                                    a = blk.reshape(E, A, B)[sel[:E]]
                                    b = dst_blk.reshape(E, A, B)[(tw[sel] - dst_rank * E)]
                                    assert np.array_equal(a, b), (leg, tt, rank, p, "twins")
                                    checks[rank] += 1
                barrier.wait()            # nobody steps on (and overwrites a buffer) before all have checked
            if rank == 0:
                times[leg] = time.perf_counter() - t_leg
            s.comm_set_overlap(False)
            for d in rs + rc:
                d.free()
        s.close()
        del ref
    except BaseException as ex:  # noqa: BLE001
        import traceback
        errors.append("rank %d: %r\n%s" % (rank, ex, traceback.format_exc()[-1500:]))
        try:
            barrier.abort()
        except Exception:  # noqa: BLE001
            pass


threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(WORLD)]
t0 = time.perf_counter()
for th in threads:
    th.start()
for th in threads:
    th.join(900)
alive = [th.is_alive() for th in threads]
print("RESULT " + json.dumps({"errors": errors, "checks": checks, "hung": alive, "world": WORLD, "envs_per_rank": E, "agents_total": WORLD * N,
                              "steps": T, "leg_seconds": times, "device_mem_used_gb": mem_used[0], "seconds": time.perf_counter() - t0}))
sys.stdout.flush()
os._exit(0 if not errors and not any(alive) else 1)
