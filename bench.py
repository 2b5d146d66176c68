#!/usr/bin/env python3
"""bench.py — agent-steps/s of the MI355X env.step() hot path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 300 --warmup 30
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one Simulator.step() of every agent on the GPU (pid + RK4 single-track dynamics,
1080-beam scan incl. noise, GJK, iTTC, opponent ray-cast, observations left in HBM), dt = 0.01 s.
Workload (BASELINE configs[2]): 65536 agents per GPU = 32768 envs x 2 agents on
example_map; env e starts on raceline waypoint (e*7919) mod 783, the opponent 10 waypoints (2 m)
behind; steer ~ U(-0.2, 0.2) / speed ~ U(2, 6) re-drawn every 20 steps; envs whose ego collides
are reset in place on the device.  Inputs (action sets, start poses, noise table) are resident in
HBM before the timed region; nothing is copied to the host inside it.

Multi-GPU: the path shards by environment (no interaction between envs), one process per GPU,
no data-path collective -> "scaling": "weak" (each rank steps its own 65536 agents).  The
rendezvous (barrier + max-over-ranks of the elapsed time) uses torch.distributed (gloo control
plane); the simulator itself never touches torch.

The JSON line also carries
  roofline      ALGORITHMIC bytes of the dominant kernel (k_scan_rays) / its HIP-event duration
  cpu_baseline  the CPU oracle (oracle/, a C port of the reference) timed on this box's host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--agents", type=int, default=65536, help="agents per GPU (envs x 2)")
    ap.add_argument("--agents-per-env", type=int, default=2)
    ap.add_argument("--beams", type=int, default=1080)
    ap.add_argument("--layout", type=int, default=int(os.environ.get("F110_MAP_LAYOUT", "3")),
                    help="0 row-major f64, 1 tiled 4x4 f64, 2 byte codes + LDS LUT, 3 row-major f64 with an out-of-bounds border + fixed-point addressing")
    ap.add_argument("--scan-block", type=int, default=int(os.environ.get("F110_SCAN_BLOCK", "0")))
    ap.add_argument("--scan-tasks", type=int, default=int(os.environ.get("F110_SCAN_TASKS", "0")),
                    help="consecutive 64-ray tasks per wave (0 = default)")
    ap.add_argument("--gather", action="store_true",
                    help="RCCL all-gather of every rank's scans after each step (BASELINE config 4; off by default)")
    ap.add_argument("--no-noise", action="store_true")
    ap.add_argument("--policy", choices=["random", "pure_pursuit", "parked"], default="random",
                    help="random: pre-drawn device-resident action sets (the default workload); pure_pursuit: the reference's "
                         "example planner (examples/waypoint_follow.py) evaluated on the device every step, closed loop; "
                         "parked: zero actions, every car stays on its start pose (SURVEY 8d fixed-pose variant, a pure scan number)")
    ap.add_argument("--no-reset", action="store_true")
    ap.add_argument("--separate-reset", action="store_true", help="re-seat finished envs with a separate launch per step instead of inside the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--map-tiles", type=int, default=1, help="tile example_map k x k (BASELINE config 5 uses 2: a 3200x3200 table)")
    ap.add_argument("--fixed-pose-steps", type=int, default=100, help="also time this many steps with all cars parked (0 = skip)")
    ap.add_argument("--secondary", type=int, default=4096, help="also time this many agents (configs[1]); 0 = skip")
    ap.add_argument("--no-profile-events", action="store_true")
    return ap.parse_args()


class Rendezvous(object):
    """barrier + max-reduce across the ranks torchrun started (gloo; no GPU tensors)."""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if "F110_BENCH_DEVICE" in os.environ:   # testing aid: several ranks on one GPU
            self.local_rank = int(os.environ["F110_BENCH_DEVICE"])
        self.dist = None
        if self.world > 1:
            import torch  # imported BEFORE libf110_hip.so so both share one libamdhip64
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            self.dist, self.torch = dist, torch

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max(self, x):
        if not self.dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def sum(self, x):
        if not self.dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t[0])

    def broadcast_bytes(self, payload, n):
        """rank 0's `payload` (n bytes) to every rank"""
        if not self.dist:
            return payload
        t = self.torch.zeros(n, dtype=self.torch.uint8)
        if self.rank == 0:
            t = self.torch.tensor(list(payload), dtype=self.torch.uint8)
        self.dist.broadcast(t, src=0)
        return bytes(t.tolist())

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


def shard_envs(total_envs_per_rank, rank):
    """global env ids owned by `rank` (contiguous blocks)"""
    import numpy as np
    return np.arange(total_envs_per_rank, dtype=np.int64) + rank * total_envs_per_rank


def start_poses_for(env_ids, num_agents, gap_wp=10):
    import numpy as np
    from _util import raceline
    w = raceline()
    n = w.shape[0]
    poses = np.empty((len(env_ids), num_agents, 3))
    for a in range(num_agents):
        k = ((env_ids * 7919) % n - a * gap_wp) % n
        poses[:, a, 0] = w[k, 1]
        poses[:, a, 1] = w[k, 2]
        poses[:, a, 2] = w[k, 3] + np.pi / 2
    return poses.reshape(len(env_ids) * num_agents, 3)


def action_sets(n_sets, n_agents, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    return [np.stack([rng.uniform(-0.2, 0.2, n_agents), rng.uniform(2.0, 6.0, n_agents)], axis=1) for _ in range(n_sets)]


def run_gpu(args, rdv, n_agents, steps, warmup, profile_events=True):
    """returns dict(elapsed_s, scan_ms_avg, dyn_ms_avg, n_reset, sim-less copies of a parity slice)"""
    import numpy as np
    from _util import load_map_image
    from f1tenth_gym_amd import BatchSim
    A = args.agents_per_env
    E = n_agents // A
    env_ids = shard_envs(E, rdv.rank)
    img, res, origin = load_map_image("example_map")
    if args.map_tiles > 1:
        # BASELINE config 5: example_map tiled k x k, resolution unchanged; the image is
        # stored top row first and the origin is the bottom-left corner, so tile (0,0) of the world
        # (where the raceline starts live) is the bottom-left copy and the origin does not move
        img = np.tile(img, (args.map_tiles, args.map_tiles))
    sim = BatchSim(num_envs=E, num_agents=A, num_beams=args.beams, device_id=rdv.local_rank,
                   map_layout=args.layout, scan_block=args.scan_block, scan_tasks_per_wave=args.scan_tasks)
    sim.set_map_image(img, res, origin)
    total = warmup + steps
    if not args.no_noise:
        sim.set_noise_table(np.random.default_rng(12345).normal(0., 0.01, size=(total + 2, args.beams)))
    poses = start_poses_for(env_ids, A)
    d_start = sim.device_array((E * A, 3)); d_start.upload(poses)
    sets = action_sets((total + 19) // 20, E * A, seed=1000 + rdv.rank)
    d_sets = []
    for s in sets:
        d = sim.device_array((E * A, 2)); d.upload(s); d_sets.append(d)
    d_count = sim.device_array((1,), dtype=np.int32); d_count.upload(np.zeros(1, dtype=np.int32))
    d_all = None
    if args.gather:
        uid = BatchSim.comm_unique_id() if rdv.rank == 0 else b"\0" * 128
        sim.comm_init(rdv.world, rdv.rank, rdv.broadcast_bytes(uid, 128))
        d_all = sim.device_array((rdv.world, E * A, args.beams))
    sim.reset_device(d_start)
    sim.sync()
    # finished envs (ego collided) are re-seated in place every step: folded into the step's last
    # kernel (f110_set_auto_reseat), or as the separate f110_reset_collided_device launch
    fused_reset = not args.no_reset and not args.separate_reset
    if fused_reset:
        sim.set_auto_reseat(d_start, 0, d_count)

    planner, d_plan = None, None
    if args.policy == "pure_pursuit":
        from f1tenth_gym_amd import PurePursuitPlanner
        from _util import raceline
        w = raceline()
        planner = PurePursuitPlanner(np.ascontiguousarray(np.stack([w[:, 1], w[:, 2], w[:, 5]], axis=1)), 0.17145 + 0.15875, sim=sim)
        d_plan = sim.device_array((E * A, 2))

    d_zero = None
    if args.policy == "parked":
        d_zero = sim.device_array((E * A, 2)); d_zero.upload(np.zeros((E * A, 2)))

    def one(t):
        if d_zero is not None:
            sim.step_device(d_zero)
        elif planner is not None:
            planner.plan_device(sim, d_plan, 0.82461887897713965, 1.375 * 0.8)   # the example's look-ahead; 80 % of its speed gain
            sim.step_device(d_plan)
        else:
            sim.step_device(d_sets[t // 20])
        if d_all is not None:
            sim.comm_all_gather_scans(d_all)
        if not args.no_reset and not fused_reset:
            sim.reset_collided_device(d_start, 0, d_count)

    for t in range(warmup):
        one(t)
    sim.sync()
    d_count.upload(np.zeros(1, dtype=np.int32))
    if profile_events:
        sim.profile_kernels(True)
    rdv.barrier()
    sim.sync()
    t0 = time.perf_counter()
    sim.timer_begin()
    for t in range(warmup, total):
        one(t)
    gpu_ms = sim.timer_end_ms()       # records + waits for the end event on the stream
    sim.sync()
    rdv.barrier()
    elapsed = time.perf_counter() - t0
    out = {"elapsed_s": elapsed, "gpu_ms": gpu_ms, "n_reset": int(d_count.download()[0]), "E": E, "A": A}
    if profile_events:
        n, scan_ms, dyn_ms, fin_ms = sim.profile_read()
        out.update({"scan_ms_avg": scan_ms / max(n, 1), "dyn_ms_avg": dyn_ms / max(n, 1),
                    "fin_ms_avg": fin_ms / max(n, 1), "n_prof": n})
        sim.profile_kernels(False)
    out["final"] = sim.get("collisions", "in_collision", "step_count")
    if d_all is not None:   # the gathered block of this rank must equal its own scans
        mine = sim.get("scans")["scans"]
        got = d_all.download()[rdv.rank]
        out["gather_ok"] = bool((mine == got).all())
        d_all.free()
    for d in d_sets + [d_start, d_count] + [x for x in (d_plan, d_zero) if x is not None]:
        d.free()
    sim.close()
    return out


def parity_gate(args, rdv):
    """first 64 envs x 200 steps of the bench inputs: HIP vs oracle (flags exact, floats <= 1e-5)"""
    import numpy as np
    from _util import load_map_image, oracle_map_dt
    from oracle import orc
    from f1tenth_gym_amd import BatchSim
    A, E, T = args.agents_per_env, 64, 200
    img, res, origin = load_map_image("example_map")
    dt, _, _ = oracle_map_dt("example_map")
    noise = None if args.no_noise else np.random.default_rng(12345).normal(0., 0.01, size=(T + 2, args.beams))
    sim = BatchSim(num_envs=E, num_agents=A, num_beams=args.beams, device_id=rdv.local_rank, map_layout=args.layout,
                   scan_block=args.scan_block, scan_tasks_per_wave=args.scan_tasks)
    sim.set_map_image(img, res, origin)
    ref = orc.SimOracle(E, A, num_beams=args.beams)
    ref.set_map_dt(dt, res, origin)
    if noise is not None:
        sim.set_noise_table(noise); ref.set_noise(noise)
    poses = start_poses_for(shard_envs(E, 0), A)
    sim.reset(poses); ref.reset(poses)
    sets = action_sets((T + 19) // 20, E * A, seed=1000)
    flag_mismatch, es, er = 0, 0.0, 0.0
    threads = min(os.cpu_count() or 1, 32)
    for t in range(T):
        sim.step(sets[t // 20]); ref.step(sets[t // 20], threads)
        if not args.no_reset:
            mask = (ref.collisions.reshape(E, A)[:, 0] != 0).astype(np.uint8)
            sim.reset(poses, mask); ref.reset(poses, mask)
        if t % 8 == 7 or t == T - 1:
            o = sim.get("scans", "state", "collisions", "in_collision")
            flag_mismatch += int(np.sum(o["collisions"] != ref.collisions) + np.sum(o["in_collision"] != ref.in_collision))
            es = max(es, float(np.max(np.abs(o["state"] - ref.state) / np.maximum(1.0, np.abs(ref.state)))))
            er = max(er, float(np.max(np.abs(o["scans"] - ref.scans) / np.maximum(1.0, np.abs(ref.scans)))))
    sim.close()
    return {"envs": E, "steps": T, "flag_mismatches": flag_mismatch, "max_rel_err_state": es, "max_rel_err_scan": er,
            "ok": bool(flag_mismatch == 0 and es < 1e-5 and er < 1e-5)}


def cpu_baseline(args, seconds):
    """time the CPU oracle (C port of the reference, OpenMP over envs) on a bounded sample of the
    same workload; also yields L-bar = mean table lookups per ray on these inputs."""
    import numpy as np
    from _util import oracle_map_dt
    from oracle import orc
    A = args.agents_per_env
    threads = max(1, min(os.cpu_count() or 1, 64))
    E = max(64, 32 * threads)
    dt, res, origin = oracle_map_dt("example_map")
    ref = orc.SimOracle(E, A, num_beams=args.beams)
    ref.set_map_dt(dt, res, origin)
    T_est = 400
    if not args.no_noise:
        ref.set_noise(np.random.default_rng(12345).normal(0., 0.01, size=(T_est + 2, args.beams)))
    poses = start_poses_for(shard_envs(E, 0), A)
    ref.reset(poses)
    sets = action_sets((T_est + 19) // 20, E * A, seed=1000)
    # calibrate, then run ~`seconds`
    t0 = time.perf_counter(); ref.step(sets[0], threads); one = time.perf_counter() - t0
    steps = int(max(3, min(T_est - 1, seconds / max(one, 1e-6))))
    look0 = ref.lookups
    t0 = time.perf_counter()
    for t in range(1, 1 + steps):
        ref.step(sets[t // 20], threads)
        if not args.no_reset:
            mask = (ref.collisions.reshape(E, A)[:, 0] != 0).astype(np.uint8)
            if mask.any():
                ref.reset(poses, mask)
    el = time.perf_counter() - t0
    lbar = (ref.lookups - look0) / float(steps * E * A * args.beams)
    return {"value": E * A * steps / el, "unit": "agent-steps/s", "cores": threads, "kind": "port",
            "sample": "%d envs x %d agents x %d steps of the bench workload (oracle/f110_oracle.c, gcc -O2 "
                      "-ffp-contract=off, OpenMP over envs), %.1f s" % (E, A, steps, el)}, lbar


def load_pmc_traffic(args, n_agents):
    """HBM bytes per scan-kernel launch from a committed rocprofv3 --pmc pass of this same
    configuration (profiles/pmc_scan.json), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_scan.json")
    if not os.path.isfile(p):
        return None
    try:
        rec = json.load(open(p))
        key = "agents=%d,beams=%d,layout=%d" % (n_agents, args.beams, args.layout)
        return rec.get(key, {}).get("hbm_bytes_per_launch")
    except Exception:  # noqa: BLE001
        return None


def main():
    args = parse_args()
    rdv = Rendezvous()
    if rdv.world != args.gpus and rdv.rank == 0 and rdv.world > 1:
        print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, rdv.world), file=sys.stderr)
    import __graft_entry__
    if rdv.local_rank == 0:
        __graft_entry__.build()
    rdv.barrier()

    res = run_gpu(args, rdv, args.agents, args.steps, args.warmup, profile_events=not args.no_profile_events)
    elapsed = rdv.max(res["elapsed_s"])
    n_gpus = rdv.world
    total_agents = args.agents * n_gpus
    value = total_agents * args.steps / elapsed
    n_reset = rdv.sum(res["n_reset"])

    line = {
        "metric": "agent-steps/s (1080-beam scan + ST dynamics)", "value": value, "unit": "agent-steps/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("%d agents per GPU (%d envs x %d), example_map%s @0.0625 m, %d-beam lidar, "
                                "ST dynamics RK4 dt=0.01, scan noise %s, in-place resets %s"
                                % (args.agents, args.agents // args.agents_per_env, args.agents_per_env,
                                   " 1600x1600" if args.map_tiles == 1 else " tiled %dx%d = %dx%d cells" % (args.map_tiles, args.map_tiles, 1600 * args.map_tiles, 1600 * args.map_tiles),
                                   args.beams,
                                   "off" if args.no_noise else "on", "off" if args.no_reset else "on"))
                               + (" (BASELINE configs[2])" if args.agents == 65536 and args.beams == 1080 else ""),
                   "policy": {"random": "pre-drawn random actions, device resident",
                              "pure_pursuit": "reference pure-pursuit planner evaluated on the device every step (closed loop, planner time included)",
                              "parked": "zero actions: every car stays on its start pose"}[args.policy],
                   "agents_per_gpu": args.agents, "agents_total": total_agents, "beams": args.beams,
                   "map_layout": {0: "rowmajor_f64", 1: "tiled4x4_f64", 2: "code8_lds_lut", 3: "padded_rowmajor_f64"}[args.layout],
                   "scan_block": args.scan_block, "scan_tasks_per_wave": args.scan_tasks,
                   "parallelism": "env-sharded x%d, %s" % (n_gpus, "RCCL all-gather of scans after every step" if args.gather
                                                           else "no data-path collective"),
                   "env_resets_in_timed_region": int(n_reset)},
    }
    if args.gather:
        line["config"]["gather_ok"] = res.get("gather_ok")
    if rdv.rank == 0:
        lbar = None
        if n_gpus == 1 and not args.no_cpu_baseline:
            cb, lbar = cpu_baseline(args, args.cpu_seconds)
            line["cpu_baseline"] = cb
            line["parity_gate"] = parity_gate(args, rdv)
        if "scan_ms_avg" in res:
            if lbar is None:
                lbar = 6.65   # SURVEY §6 probe value; replaced by the measured one whenever the CPU leg runs
            B = args.beams
            scan_bytes = args.agents * (8.0 * B + 8.0 * B * lbar)   # range write + L-bar gathers of 8 B per ray
            step_bytes = args.agents * ((216.0 + 8.0 * B) + 8.0 * B * lbar)  # SURVEY §8d B_alg
            ach = scan_bytes / (res["scan_ms_avg"] * 1e-3) / 1e9
            line["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": ach / HBM_PEAK_GBS, "traffic": load_pmc_traffic(args, args.agents),
                                "kernel": "k_scan_rays_agent" if (args.layout == 3 and args.beams < 1498 and (-args.beams) % 64 * 100 <= 3 * args.beams) else "k_scan_rays", "kernel_ms_avg": res["scan_ms_avg"],
                                "integrate_collide_ms_avg": res["dyn_ms_avg"], "finalize_ms_avg": res["fin_ms_avg"],
                                "launches_timed": res["n_prof"],
                                "alg_bytes_per_launch": scan_bytes, "lookups_per_ray": lbar,
                                **({"note": "more beams than table directions (theta_dis = 2000): beams that share a direction are "
                                            "marched once and expanded, so the algorithmic bytes (counted per beam, as the "
                                            "reference works) can exceed what the kernel had to fetch and frac can pass 1"}
                                   if B >= 1498 else {}),
                                "step_alg_bytes": step_bytes,
                                "step_achieved_GBs": step_bytes * args.steps / elapsed / 1e9}
    if n_gpus == 1 and args.secondary and args.secondary != args.agents and rdv.rank == 0:
        r2 = run_gpu(args, rdv, args.secondary, max(args.steps, 200), args.warmup, profile_events=False)
        line["config"]["secondary"] = {"workload": "%d agents (BASELINE configs[1])" % args.secondary,
                                       "value": args.secondary * max(args.steps, 200) / r2["elapsed_s"],
                                       "ms_per_step": 1e3 * r2["elapsed_s"] / max(args.steps, 200)}
    if n_gpus == 1 and args.fixed_pose_steps > 0 and args.policy == "random" and rdv.rank == 0:
        import copy
        a3 = copy.copy(args)
        a3.policy, a3.no_reset = "parked", True
        r3 = run_gpu(a3, rdv, args.agents, args.fixed_pose_steps, 10, profile_events=False)
        line["config"]["fixed_pose_variant"] = {"workload": "same agents parked on their start poses (speed 0, no resets)",
                                                "value": args.agents * args.fixed_pose_steps / r3["elapsed_s"],
                                                "ms_per_step": 1e3 * r3["elapsed_s"] / args.fixed_pose_steps}
    if rdv.rank == 0:
        print(json.dumps(line))
        sys.stdout.flush()
    rdv.barrier()
    rdv.close()


if __name__ == "__main__":
    main()
