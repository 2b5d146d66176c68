"""CPU-only, world_size 2 over gloo: bench.py's rendezvous (barrier, max / sum over ranks) and the
env sharding used for N > 1 GPUs (contiguous env blocks, disjoint start poses, per-rank seeds)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import bench
rdv = bench.Rendezvous()
rdv.barrier()
ids = bench.shard_envs(5, rdv.rank)
poses = bench.start_poses_for(ids, 2)
acts = bench.action_sets(2, 10, seed=1000 + rdv.rank)
out = {"rank": rdv.rank, "world": rdv.world, "max": rdv.max(1.0 + rdv.rank), "sum": rdv.sum(10.0 * (rdv.rank + 1)),
       "ids": ids.tolist(), "pose0": poses[0].tolist(), "act0": acts[0][0].tolist()}
rdv.barrier()
print("RESULT " + json.dumps(out)); sys.stdout.flush()
rdv.close()
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_rank_rendezvous_and_sharding():
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT}], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=240)
        assert p.returncode == 0, se[-2000:]
        outs.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][0][7:]))
    outs.sort(key=lambda o: o["rank"])
    assert [o["world"] for o in outs] == [2, 2]
    assert outs[0]["max"] == outs[1]["max"] == 2.0          # max over ranks (the timing rule)
    assert outs[0]["sum"] == outs[1]["sum"] == 30.0
    assert outs[0]["ids"] == [0, 1, 2, 3, 4] and outs[1]["ids"] == [5, 6, 7, 8, 9]   # contiguous, disjoint
    assert outs[0]["pose0"] != outs[1]["pose0"] and outs[0]["act0"] != outs[1]["act0"]


def test_single_process_rendezvous_is_a_noop():
    sys.path.insert(0, ROOT)
    import bench
    env = {k: os.environ.pop(k) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK") if k in os.environ}
    try:
        r = bench.Rendezvous()
        assert r.world == 1 and r.max(3.5) == 3.5 and r.sum(2.0) == 2.0
        r.barrier(); r.close()
    finally:
        os.environ.update(env)
    ids = bench.shard_envs(4, 3)
    assert ids.tolist() == [12, 13, 14, 15]
    p = bench.start_poses_for(ids, 2)
    from _util import bench_start_poses
    assert np.array_equal(p, bench_start_poses(16, 2)[24:32])
