#!/usr/bin/env python3
"""E environments over several MI355X from one process: `ShardedVecEnv` — contiguous env blocks, one handle and one worker thread per
device, no collective on the step path (the reference's serial agent loop, base_classes.py:568-585, is what gets sharded).

    python examples/sharded_vec_env.py --envs 131072 --devices 0,1,2,3,4,5,6,7      # 262 144 agents over the 8 GPUs of a node
    python examples/sharded_vec_env.py --envs 4096 --devices 0,0                     # two handles on one GPU (how a 1-GPU box runs it)
    python examples/sharded_vec_env.py --envs 4096 --devices 0,0 --gather            # + every shard's observation on every device (RCCL)

Host actions in, `done` + lap bookkeeping out (episode logic and auto-reset on the devices); the scans stay in HBM
(`env.device_views()[k]['scans']` on shard k's GPU) unless --scans asks for them."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f1tenth_gym_amd as amd  # noqa: E402
from f1tenth_gym_amd import workload  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--agents", type=int, default=2)
    ap.add_argument("--devices", default="0", help="comma-separated HIP device ids, one handle each (an id may repeat)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--gather", action="store_true", help="all-gather the observation over the shards after every step")
    ap.add_argument("--scans", action="store_true", help="also bring the scans to the host every step (8.6 KB per agent)")
    ap.add_argument("--threaded", type=int, default=-1, help="1 / 0: step the shards from one worker thread each / enqueue all then wait all (default: the class decides)")
    args = ap.parse_args(argv)
    E, A = args.envs, args.agents
    devices = [int(d) for d in args.devices.split(",")]
    fields = ("poses_x", "poses_y", "collisions") + (("scans",) if args.scans else ())
    env = amd.ShardedVecEnv(E, devices=devices, gather_obs=args.gather, map=workload.map_stem("example_map"), map_ext=".png", num_agents=A,
                            auto_reset=True, obs_fields=fields, threaded_step=None if args.threaded < 0 else bool(args.threaded))
    obs, _, done, info = env.reset(workload.bench_start_poses(E, A).reshape(E, A, 3))
    rng = np.random.default_rng(0)
    act = np.stack([rng.uniform(-0.2, 0.2, (E, A)), rng.uniform(2.0, 6.0, (E, A))], axis=2)
    for _ in range(20):
        env.step(act)
    n_done = 0
    t0 = time.perf_counter()
    for t in range(args.steps):
        if t % 20 == 0:
            act = np.stack([rng.uniform(-0.2, 0.2, (E, A)), rng.uniform(2.0, 6.0, (E, A))], axis=2)
        obs, _, done, info = env.step(act)
        n_done += int(done.sum())
    env.sync()
    dt = (time.perf_counter() - t0) / args.steps
    print("%d envs x %d agents over %d handle(s) on devices %s (%s): %.3f ms per step, %.1f M agent-steps/s; %d episodes ended and were re-seated%s"
          % (E, A, len(devices), devices, "one worker thread per shard" if env.threaded_step else "enqueue all, then wait all", dt * 1e3, E * A / dt / 1e6, n_done,
             "; observation gathered to every device" if args.gather else ""))
    if args.gather:
        d_scans, d_scal = env.gathered_views()[0]
        print("shard 0 holds scans %s and scalars %s of every shard" % (d_scans.shape, d_scal.shape))
    env.close()
    return E * A / dt


if __name__ == "__main__":
    main()
