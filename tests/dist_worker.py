"""One rank of the N-rank observation-gather test (launched by tests/test_gpu_multi.py, one process per
rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT in the environment as the driver's launcher sets them).

Every rank steps its own env shard, gathers the whole observation (scans + the 7 scalars per agent) with
f110_comm_all_gather_obs, and then checks EVERY OTHER rank's gathered block bit for bit against a local
re-simulation of that rank's shard (the path is deterministic: same poses, same action stream, same
noise stream -> same bits).  Two passes: the gather in the step's stream, and overlapped with the next
step (double-buffered observation) where the block of step t is read while step t+1 runs.

F110_BENCH_DEVICE=<d> puts every rank on device d (two ranks on one GPU: RCCL may refuse that).
Prints one line: RESULT {json}.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import bench  # noqa: E402
from _util import load_map_image  # noqa: E402


def make_sim(device, E, A, rank):
    from f1tenth_gym_amd import BatchSim
    img, res, origin = load_map_image("example_map")
    s = BatchSim(num_envs=E, num_agents=A, device_id=device)
    s.set_map_image(img, res, origin)
    s.set_noise_rng(12345, 0.01)
    s.reset(bench.start_poses_for(bench.shard_envs(E, rank), A))
    return s


def actions_of(rank, t, n):
    rng = np.random.default_rng(1000 * (rank + 1) + t)
    return np.stack([rng.uniform(-0.3, 0.3, n), rng.uniform(1.0, 7.0, n)], axis=1)


def observation(sim):
    o = sim.get("scans", "poses_x", "poses_y", "poses_theta", "linear_vels_x", "ang_vels_z", "collisions")
    scal = np.stack([o["poses_x"], o["poses_y"], o["poses_theta"], o["linear_vels_x"], np.zeros(sim.N), o["ang_vels_z"], o["collisions"]])
    return o["scans"], scal


def main():
    E, A, T = int(os.environ.get("F110_DIST_ENVS", "48")), 2, int(os.environ.get("F110_DIST_STEPS", "24"))
    rdv = bench.Rendezvous()
    from f1tenth_gym_amd import BatchSim, _ffi
    dev = rdv.local_rank
    N = E * A
    out = {"rank": rdv.rank, "world": rdv.world, "device": dev, "devices_visible": _ffi.device_count()}
    sim = make_sim(dev, E, A, rdv.rank)
    uid = BatchSim.comm_unique_id() if rdv.rank == 0 else b"\0" * 128
    try:
        sim.comm_init(rdv.world, rdv.rank, rdv.broadcast_bytes(uid, 128))
    except Exception as ex:  # noqa: BLE001 — e.g. two ranks on one device
        print("RESULT " + json.dumps(dict(out, rccl_refused=str(ex)[:300])))
        sys.stdout.flush()
        rdv.barrier()   # keep the peers' sockets alive until every rank has reported
        return 0
    out["rccl_ranks"], out["rccl_rank"] = sim.comm_info()
    recv = [(sim.device_array((rdv.world, N, 1080)), sim.device_array((rdv.world, 7, N))) for _ in range(2)]

    # what every OTHER rank must have produced: re-simulate its shard here, keeping each step's observation
    others = {}
    for r in range(rdv.world):
        if r == rdv.rank:
            continue
        ref = make_sim(dev, E, A, r)
        obs = []
        for t in range(2 * T):
            ref.step(actions_of(r, t, N))
            obs.append(observation(ref))
        ref.close()
        others[r] = obs

    def equal(slot, r, t):
        s, c = recv[slot][0].download()[r], recv[slot][1].download()[r]
        return bool(np.array_equal(s, others[r][t][0]) and np.array_equal(c, others[r][t][1]))

    # pass 1: gather on the step's stream, checked after every step
    bad = 0
    for t in range(T):
        sim.step(actions_of(rdv.rank, t, N))
        sim.comm_all_gather_obs(*recv[0])
        mine_s, mine_c = observation(sim)
        if not (np.array_equal(recv[0][0].download()[rdv.rank], mine_s) and np.array_equal(recv[0][1].download()[rdv.rank], mine_c)):
            bad += 1
        bad += sum(0 if equal(0, r, t) else 1 for r in others)
    out["in_stream_mismatches"] = bad

    # pass 2: overlapped — the gather of step t is consumed while step t + 1 (and its gather) are in flight
    sim.comm_set_overlap(True)
    bad = 0
    for t in range(T, 2 * T):
        sim.step(actions_of(rdv.rank, t, N))
        sim.comm_all_gather_obs(*recv[t % 2])
        if t > T:
            bad += sum(0 if equal((t - 1) % 2, r, t - 1) else 1 for r in others)
    bad += sum(0 if equal((2 * T - 1) % 2, r, 2 * T - 1) else 1 for r in others)
    sim.comm_set_overlap(False)
    out["overlapped_mismatches"] = bad
    out["steps"] = 2 * T
    rdv.barrier()
    for a, b in recv:
        a.free(); b.free()
    sim.close()
    print("RESULT " + json.dumps(out))
    sys.stdout.flush()
    rdv.barrier()
    rdv.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
