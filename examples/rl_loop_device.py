#!/usr/bin/env python3
"""A device-resident RL-style loop: observe -> act -> step with NOTHING crossing PCIe and no host synchronisation inside it.

    scans [N][1080] (HBM)  --policy-->  actions [N][2] (HBM)  --f110_episode_step_device-->  scans ...
                                                              --f110_episode_reset_done_device (auto-reset of finished envs)

Two policies:
  * the built-in stand-in `BatchSim.scan_policy_device` (a reactive "steer to the most open sector" kernel that reads every scan
    where the scan kernel left it) — no dependency beyond this package;
  * --torch: a tiny random MLP in PyTorch-ROCm fed through DLPack (`torch.from_dlpack(views["scans"])`, zero copy, no type of this
    package on torch's side), running on the simulator's own stream (`torch.cuda.ExternalStream(views["stream"])`) and writing the
    action buffer in place.

    python examples/rl_loop_device.py [--envs 4096] [--steps 500] [--torch]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--torch" in sys.argv:
    import torch  # noqa: F401  BEFORE the simulator's library: both then share one libamdhip64.so
import f1tenth_gym_amd as amd  # noqa: E402


def start_poses(E, A):
    from f1tenth_gym_amd import workload
    return workload.bench_start_poses(E, A)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--agents", type=int, default=2)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--torch", action="store_true")
    args = ap.parse_args(argv)
    E, A = args.envs, args.agents
    N = E * A
    sim = amd.BatchSim(num_envs=E, num_agents=A)
    from f1tenth_gym_amd import workload
    sim.set_map(workload.map_stem("example_map") + ".yaml", ".png")
    sim.set_noise_rng(12345, 0.01)
    sim.episode_init(0)
    sim.episode_reset(start_poses(E, A))
    actions = sim.device_array((N, 2))
    actions.upload(np.zeros((N, 2)))
    views = sim.device_views()
    ep = sim.episode_device_views()
    d_resets = sim.device_array((1,), np.int32)
    d_resets.upload(np.zeros(1, np.int32))

    policy = None
    if args.torch:
        import torch
        stream = torch.cuda.ExternalStream(views["stream"], device=torch.device("cuda", sim.device_id))   # torch enqueues behind the simulator's kernels: no event, no sync
        scans_t = torch.from_dlpack(views["scans"])                # [N][1080] float64, the simulator's own buffer
        act_t = torch.from_dlpack(actions)                         # [N][2], the buffer f110_episode_step_device reads
        g = torch.Generator(device="cuda").manual_seed(0)
        w1 = torch.randn(108, 32, device="cuda", generator=g, dtype=torch.float32) * 0.1
        w2 = torch.randn(32, 2, device="cuda", generator=g, dtype=torch.float32) * 0.1

        def policy():
            # back-to-back device steps go out as TWO env blocks on two streams (include/f110.h step_groups); torch sees only
            # the main one.  The fence orders both blocks in front of torch's reads and the next step behind torch's writes.
            sim.fence()
            with torch.cuda.stream(stream):
                x = scans_t[:, ::10].to(torch.float32).clamp_(max=10.0) / 10.0
                out = torch.tanh(torch.tanh(x @ w1) @ w2)
                act_t[:, 0] = (0.4 * out[:, 0]).to(torch.float64)
                act_t[:, 1] = (3.5 + 2.5 * out[:, 1]).to(torch.float64)
    else:
        def policy():
            sim.scan_policy_device(actions)

    sim.episode_step_device(actions)          # the first observation
    t0 = time.perf_counter()
    for _ in range(args.steps):
        policy()                              # scans (HBM) -> actions (HBM)
        sim.episode_step_device(actions)      # integrate, scan, collisions, lap logic, done
        sim.episode_reset_done_device(d_resets)   # finished envs back to their start poses, on the device
    sim.sync()                                # the only synchronisation: to read the clock
    dt = time.perf_counter() - t0
    done_now = int(ep["done"].download().sum())
    laps = ep["lap_counts"].download()
    print("%d envs x %d agents, %d steps: %.3f ms per step, %.1f M agent-steps/s; %d env resets, %d envs done right now, max lap count %.0f (%s policy)"
          % (E, A, args.steps, dt / args.steps * 1e3, N * args.steps / dt / 1e6, int(d_resets.download()[0]), done_now, laps.max(),
             "torch MLP via DLPack" if args.torch else "built-in scan"))
    if args.torch:
        # the tensors made by from_dlpack view the simulator's memory: they go first (BatchSim.close() refuses while they live)
        del scans_t, act_t, policy
        torch.cuda.synchronize()
    sim.close()
    return N * args.steps / dt


if __name__ == "__main__":
    main()
