#!/usr/bin/env python3
"""bench.py — agent-steps/s of the MI355X env.step() hot path (BASELINE.json metric).

    python bench.py                                   # 1 GPU, default K/W, all legs (a few minutes)
    python bench.py --gpus N --steps K --warmup W      # spawns N ranks itself, one per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W     # or is launched as N ranks

One "step" = one Simulator.step() of every agent on the GPU (pid + RK4 single-track dynamics,
1080-beam scan incl. noise, GJK, iTTC, opponent ray-cast, observations left in HBM), dt = 0.01 s.
Workload (BASELINE configs[2]): 65536 agents per GPU = 32768 envs x 2 agents on
example_map; env e starts on raceline waypoint (e*7919) mod 783, the opponent 10 waypoints (2 m)
behind; steer ~ U(-0.2, 0.2) / speed ~ U(2, 6) re-drawn every 20 steps; envs whose ego collides
are reset in place on the device.  Inputs (action sets, start poses) are resident in HBM before
the timed region; the scan noise is drawn on the device (NumPy's seed-12345 stream, bit for bit);
nothing is copied to the host inside the timed region.

The timed window is steady state by construction: after the reset the script runs --preroll (500)
un-timed steps of its own, then the caller's --warmup, then times exactly --steps — so a short
`--steps 20 --warmup 5` measures the regime with episodes of every age and in-place resets, not the
easy steps right after a reset (`steady_state.headline_over_steady` checks it against the >= 1000-step leg).

Multi-GPU: the path shards by environment (no interaction between envs), one process per GPU,
no data-path collective -> "scaling": "weak" (each rank steps its own 65536 agents).  The
control plane (barrier + max-over-ranks of the elapsed time) is a few lines of stdlib sockets
(class Rendezvous): no torch anywhere.  Launched without a rank environment and with --gpus N > 1
the script spawns the N ranks itself and fails loudly when fewer than N devices are visible.
Each rank pins itself to its GPU's NUMA node (f1tenth_gym_amd/numa.py).  With N > 1 ONE invocation
reports three legs (`multi_gpu`): the headline without any collective, the same steps with the RCCL
observation gather (scans + 7 scalars per agent, f110_comm_all_gather_obs) on the step's stream, and
with the gather overlapped with the next step — each with every rank's own ms per step (min / max),
the communicator size as RCCL reports it, and the per-GPU rate; at --gpus 8 also BASELINE configs[3] to the
letter (262144 agents over the node, without / with the gather).  The gather legs run last, under a watchdog:
whatever RCCL does on the node, the line with the headline is printed.

The JSON line carries, besides the contract's fields:
  roofline      SURVEY §8d: whole-step ALGORITHMIC bytes (216 + 8B + 8B*L-bar per agent-step) x
                agent-steps/s over 8 TB/s as `frac`; the scan kernel's own figure (HIP events, a
                separate replay of the same steps) as `kernel_frac`; L-bar counted ON THE DEVICE
                over exactly the timed steps (a third replay with the counting kernel variant);
                the compulsory-stream fraction, the PMC-measured HBM fraction, and the binding
                gather-issue floor (profiles/rNN_issue_floor.json of the library's sources)
  steady_state  the same workload over >= 1000 timed steps after >= 100 warm-up steps
  cpu_baseline  the CPU oracle (oracle/, a C port of the reference) on this box's host cores, plus
                `cpu_1t`: the reference's own shape (1 env x 2 agents, one thread; BASELINE configs[0])
  config.secondary / config.config5 / config.fixed_pose_variant: the other BASELINE configs
"""
import argparse
import json
import os
import socket
import struct
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)   # (nothing under tests/ is imported: the workload lives in f1tenth_gym_amd/workload.py)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
NOISE_SEED = 12345      # F110Env's default seed (f110_env.py:107)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--preroll", type=int, default=500,
                    help="un-timed steps between the reset and the warm-up: the batch reaches its steady regime (episodes of every age, "
                         "in-place resets, long rays) before anything is timed, whatever --warmup / --steps the caller picks")
    ap.add_argument("--agents", type=int, default=65536, help="agents per GPU (envs x 2)")
    ap.add_argument("--agents-per-env", type=int, default=2)
    ap.add_argument("--beams", type=int, default=1080)
    ap.add_argument("--layout", type=int, default=int(os.environ.get("F110_MAP_LAYOUT", "3")),
                    help="0 row-major f64, 1 tiled 4x4 f64, 2 byte codes + LDS LUT, 3 row-major f64 with an out-of-bounds border + fixed-point addressing, 4 = 3 + the 128x128 cells around each lidar staged in LDS")
    ap.add_argument("--scan-block", type=int, default=int(os.environ.get("F110_SCAN_BLOCK", "0")))
    ap.add_argument("--scan-tasks", type=int, default=int(os.environ.get("F110_SCAN_TASKS", "0")),
                    help="consecutive 64-ray tasks per wave (0 = default)")
    ap.add_argument("--groups", type=int, default=0, help="env groups stepped on streams of their own (0 = the library's default)")
    ap.add_argument("--gather", action="store_true",
                    help="RCCL all-gather of every rank's scans after each step (BASELINE config 4; off by default)")
    ap.add_argument("--gather-overlap", action="store_true",
                    help="with --gather: double-buffered scans, the gather of step t runs beside step t+1 (f110_comm_set_overlap)")
    ap.add_argument("--no-gather-legs", action="store_true",
                    help="N > 1: skip the two extra legs (observation gather in-stream, and overlapped) that follow the headline")
    ap.add_argument("--gather-legs", action="store_true", help="run the gather legs at N = 1 too (a device copy there)")
    ap.add_argument("--no-numa", action="store_true", help="do not pin the rank to its GPU's NUMA node")
    ap.add_argument("--config3-legs", action="store_true",
                    help="also run BASELINE configs[3] to the letter (262144 agents over the node, without / with the gather); automatic at --gpus 8")
    ap.add_argument("--gather-timeout", type=float, default=120.0, help="watchdog (s) per gather leg: communicator init + the leg's steps")
    ap.add_argument("--gather-budget", type=float, default=240.0, help="seconds all gather legs of one invocation may take together: legs that would start later are skipped and listed")
    ap.add_argument("--noise", choices=["rng", "table", "off"], default="rng",
                    help="rng: drawn on the device (default); table: NumPy's rows uploaded (A/B); off")
    ap.add_argument("--no-noise", action="store_true", help="same as --noise off")
    ap.add_argument("--policy", choices=["random", "pure_pursuit", "parked"], default="random",
                    help="random: pre-drawn device-resident action sets (the default workload); pure_pursuit: the reference's "
                         "example planner (examples/waypoint_follow.py) evaluated on the device every step, closed loop; "
                         "parked: zero actions, every car stays on its start pose (SURVEY 8d fixed-pose variant, a pure scan number)")
    ap.add_argument("--no-reset", action="store_true")
    ap.add_argument("--separate-reset", action="store_true", help="re-seat finished envs with a separate launch per step instead of inside the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in legs (F110Env on 1 env, F110VecEnv at 2048 / 32768 envs)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--map-tiles", type=int, default=1, help="tile example_map k x k (BASELINE config 5 uses 2: a 3200x3200 table)")
    ap.add_argument("--fixed-pose-steps", type=int, default=100, help="also time this many steps with all cars parked (0 = skip)")
    ap.add_argument("--secondary", type=int, default=4096, help="also time this many agents (configs[1]); 0 = skip")
    ap.add_argument("--steady-steps", type=int, default=1000, help="steady-state leg: timed steps (SURVEY 8d: >= 1000); 0 = skip")
    ap.add_argument("--steady-warmup", type=int, default=100)
    ap.add_argument("--no-config5", action="store_true", help="skip the 65536 x 4096-beam / 3200x3200-table leg")
    ap.add_argument("--only-headline", action="store_true", help="the timed region only (profiling / PMC runs): no replays, no other legs")
    ap.add_argument("--no-profile-events", action="store_true")
    ap.add_argument("--ranks-in-process", action="store_true",
                    help="the N ranks of --gpus N are THREADS of this one process, each with its own handle (SURVEY 8e's \"one process with 8 "
                         "handles\"): rank r drives device r, or every rank the device F110_BENCH_DEVICE names.  The same legs, control plane, "
                         "digests and records as one process per GPU (the contract's launch form, and the default); no NUMA pinning per rank")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)   # tests: no GPU, a sleep stands in for the step
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# control plane: barrier / max / sum / broadcast over a Unix socket, rank 0 is the hub
class Rendezvous(object):
    """The ranks of ONE node (torchrun's or our own): RANK / LOCAL_RANK / WORLD_SIZE from the
    environment, the meeting point derived from F110_BENCH_RDV or MASTER_PORT.  Every collective
    is: each rank sends its value to rank 0, rank 0 replies with the reduction."""

    def __init__(self, ident=None):
        """ident: None = this process is one rank, identity from the launcher's environment; (world, rank, local_rank, name) =
        a rank that is a THREAD of this process (--ranks-in-process)"""
        if ident is None:
            self.world = int(os.environ.get("WORLD_SIZE", "1"))
            self.rank = int(os.environ.get("RANK", "0"))
            self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
            name = os.environ.get("F110_BENCH_RDV") or ("f110-bench-%s-%s" % (os.environ.get("MASTER_PORT", "0"),
                                                                            os.environ.get("TORCHELASTIC_RUN_ID", "none")))
            self.in_process = False
        else:
            self.world, self.rank, self.local_rank, name = ident
            self.in_process = True
        if "F110_BENCH_DEVICE" in os.environ:   # testing aid: several ranks on one GPU
            self.local_rank = int(os.environ["F110_BENCH_DEVICE"])
        self.peers, self.sock = [], None
        if self.world > 1:
            addr = "\0" + name    # abstract namespace: nothing to clean up, nothing on disk
            if self.rank == 0:
                srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                srv.bind(addr)
                srv.listen(self.world)
                srv.settimeout(300)
                got = {}
                while len(got) < self.world - 1:
                    c, _ = srv.accept()
                    c.settimeout(600)
                    r = struct.unpack("<i", self._recv(c, 4))[0]
                    got[r] = c
                self.peers = [got[r] for r in sorted(got)]
                srv.close()
            else:
                deadline = time.time() + 300
                while True:
                    try:
                        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                        s.connect(addr)
                        break
                    except (ConnectionRefusedError, FileNotFoundError):
                        s.close()
                        if time.time() > deadline:
                            raise RuntimeError("bench rendezvous: rank 0 never appeared at %r" % name)
                        time.sleep(0.05)
                s.settimeout(600)
                s.sendall(struct.pack("<i", self.rank))
                self.sock = s

    @staticmethod
    def _recv(c, n):
        buf = b""
        while len(buf) < n:
            part = c.recv(n - len(buf))
            if not part:
                raise RuntimeError("bench rendezvous: peer went away")
            buf += part
        return buf

    def _reduce(self, payload, combine):
        """payload: bytes of fixed length; combine(list_of_payloads) -> bytes; everybody gets the result"""
        if self.world == 1:
            return combine([payload])
        if self.rank == 0:
            vals = [payload] + [self._recv(c, len(payload)) for c in self.peers]
            out = combine(vals)
            for c in self.peers:
                c.sendall(out)
            return out
        self.sock.sendall(payload)
        return self._recv(self.sock, len(payload))

    def barrier(self):
        self._reduce(b"\0", lambda v: b"\0")

    def max(self, x):
        return struct.unpack("<d", self._reduce(struct.pack("<d", x), lambda v: struct.pack("<d", max(struct.unpack("<d", b)[0] for b in v))))[0]

    def sum(self, x):
        return struct.unpack("<d", self._reduce(struct.pack("<d", x), lambda v: struct.pack("<d", sum(struct.unpack("<d", b)[0] for b in v))))[0]

    def gather_bytes(self, payload):
        """every rank's fixed-length payload, in rank order, on every rank"""
        n = len(payload)
        if self.world == 1:
            return [bytes(payload)]
        if self.rank == 0:
            vals = [bytes(payload)] + [self._recv(c, n) for c in self.peers]
            out = b"".join(vals)
            for c in self.peers:
                c.sendall(out)
        else:
            self.sock.sendall(bytes(payload))
            out = self._recv(self.sock, n * self.world)
        return [out[i * n:(i + 1) * n] for i in range(self.world)]

    def gather(self, x):
        return [struct.unpack("<d", b)[0] for b in self.gather_bytes(struct.pack("<d", x))]

    def broadcast_bytes(self, payload, n):
        """rank 0's `payload` (n bytes) to every rank"""
        mine = bytes(payload) if self.rank == 0 else b"\0" * n
        return self._reduce(mine, lambda v: v[0])

    def close(self):
        for c in self.peers:
            c.close()
        if self.sock:
            self.sock.close()


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` outside any launcher: start the N ranks (one process per GPU of this
    node), stream rank 0's stdout through, fail if any rank fails.  Returns the exit code."""
    n = args.gpus
    if not args.stub:
        from f1tenth_gym_amd import _ffi
        import __graft_entry__
        __graft_entry__.build()
        have = _ffi.device_count()
        if have < n and "F110_BENCH_DEVICE" not in os.environ:   # (the testing aid puts every rank on one named device)
            print("bench.py: --gpus %d but only %d HIP device(s) visible — refusing to run a smaller job under that label" % (n, have),
                  file=sys.stderr)
            return 2
    rdv = "f110-bench-self-%d-%d" % (os.getpid(), int(time.time() * 1e3) % 100000)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), F110_BENCH_RDV=rdv)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = rc or p.wait()
    return rc


def thread_ranks(args, argv):
    """--ranks-in-process: the N ranks as threads of this process, one handle each (ctypes calls release the GIL; a step is
    a few microseconds of Python).  Same main() per rank; rank 0 prints the line.  Returns the exit code."""
    n = args.gpus
    if not args.stub:
        from f1tenth_gym_amd import _ffi
        import __graft_entry__
        __graft_entry__.build()
        have = _ffi.device_count()
        if have < n and "F110_BENCH_DEVICE" not in os.environ:
            print("bench.py: --gpus %d --ranks-in-process but only %d HIP device(s) visible — refusing to run a smaller job under that label" % (n, have),
                  file=sys.stderr)
            return 2
    name = "f110-bench-threads-%d-%d" % (os.getpid(), int(time.time() * 1e3) % 100000)
    rcs = [1] * n

    def run(r):
        try:
            rcs[r] = main(argv, ident=(n, r, r, name))
        except SystemExit as ex:
            rcs[r] = int(ex.code or 0) if not isinstance(ex.code, str) else 1
            if isinstance(ex.code, str):
                print(ex.code, file=sys.stderr)
        except BaseException:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            rcs[r] = 1
            os._exit(1)   # the other ranks would wait for this one at the next barrier
    ths = [threading.Thread(target=run, args=(r,), name="rank%d" % r) for r in range(n)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    return max(rcs)


# ---------------------------------------------------------------------------------------------
def shard_envs(total_envs_per_rank, rank):
    """global env ids owned by `rank` (contiguous blocks)"""
    from f1tenth_gym_amd import workload
    return workload.shard_envs(total_envs_per_rank, rank)


def start_poses_for(env_ids, num_agents, gap_wp=10):
    from f1tenth_gym_amd import workload
    # F110_BENCH_START_ORDER: experiment only (tools/debug): "sorted" lays the envs out along the track
    return workload.start_poses(env_ids, num_agents, gap_wp, order=os.environ.get("F110_BENCH_START_ORDER", ""))


def action_sets(n_sets, n_agents, seed):
    from f1tenth_gym_amd import workload
    return workload.action_sets(n_sets, n_agents, seed)


def rel_err(a, b, atol=1e-12):
    """the smallest tol with |a - b| <= tol*|b| + atol everywhere (relative to the reference value itself)"""
    import numpy as np
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if not a.size:
        return 0.0
    excess = np.maximum(np.abs(a - b) - atol, 0.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        e = np.where(excess > 0.0, excess / np.abs(b), 0.0)
    return float(np.max(e))


def oracle_map_dt(name="example_map"):
    """the example track's distance table by the ORACLE's EDT (checker side only: parity_gate / cpu_baseline)"""
    from oracle import orc
    from f1tenth_gym_amd.workload import load_map_image
    img, res, origin = load_map_image(name)
    return orc.map_dt_from_image(img, res), res, origin


class Workload(object):
    """One rank's simulator + device-resident inputs.  run() always starts from the same reset, so
    passes over the same (warmup, steps) replay the same states: the timed pass is clean, the
    per-kernel events and the lookup count come from replays of the identical steps."""

    def __init__(self, args, rdv, n_agents, max_total_steps, beams=None, map_tiles=None, policy=None, no_reset=None):
        import numpy as np
        from f1tenth_gym_amd import BatchSim
        from f1tenth_gym_amd.workload import load_map_image
        self.args, self.rdv, self.np = args, rdv, np
        self.beams = args.beams if beams is None else beams
        self.tiles = args.map_tiles if map_tiles is None else map_tiles
        self.policy = args.policy if policy is None else policy
        self.no_reset = args.no_reset if no_reset is None else no_reset
        A = args.agents_per_env
        self.A, self.E, self.N = A, n_agents // A, n_agents
        env_ids = shard_envs(self.E, rdv.rank)
        img, res, origin = load_map_image("example_map")
        if self.tiles > 1:
            # BASELINE config 5: example_map tiled k x k, resolution unchanged; the image is
            # stored top row first and the origin is the bottom-left corner, so tile (0,0) of the world
            # (where the raceline starts live) is the bottom-left copy and the origin does not move
            img = np.tile(img, (self.tiles, self.tiles))
        self.sim = sim = BatchSim(num_envs=self.E, num_agents=A, num_beams=self.beams, device_id=rdv.local_rank,
                                  map_layout=args.layout, scan_block=args.scan_block, scan_tasks_per_wave=args.scan_tasks,
                                  step_groups=args.groups)
        sim.set_map_image(img, res, origin)
        self.max_total = max_total_steps
        noise = "off" if args.no_noise else args.noise
        if noise == "rng":
            sim.set_noise_rng(NOISE_SEED, 0.01)
            sim.noise_prepare(max_total_steps + 2)   # row cache filled before any timed region
        elif noise == "table":
            sim.set_noise_table(np.random.default_rng(NOISE_SEED).normal(0., 0.01, size=(max_total_steps + 2, self.beams)))
        self.noise = noise
        self.poses = start_poses_for(env_ids, A)
        self.d_start = sim.device_array((self.N, 3)); self.d_start.upload(self.poses)
        self.d_sets = []
        for s in action_sets((max_total_steps + 19) // 20, self.N, seed=1000 + rdv.rank):
            d = sim.device_array((self.N, 2)); d.upload(s); self.d_sets.append(d)
        self.d_count = sim.device_array((1,), dtype=np.int32)
        self.d_all = None       # gather off
        self.d_alls, self.d_scals, self.comm_ready = [], [], False
        self.g_f32, self.g_root = False, None
        if args.gather:
            self.set_gather(True, args.gather_overlap)
        self.planner = self.d_plan = self.d_zero = None
        if self.policy == "pure_pursuit":
            from f1tenth_gym_amd import PurePursuitPlanner
            from f1tenth_gym_amd.workload import raceline
            w = raceline()
            self.planner = PurePursuitPlanner(np.ascontiguousarray(np.stack([w[:, 1], w[:, 2], w[:, 5]], axis=1)), 0.17145 + 0.15875, sim=sim)
            self.d_plan = sim.device_array((self.N, 2))
        if self.policy == "parked":
            self.d_zero = sim.device_array((self.N, 2)); self.d_zero.upload(np.zeros((self.N, 2)))
        # finished envs (ego collided) are re-seated in place every step: folded into the step's last
        # kernel (f110_set_auto_reseat), or as the separate f110_reset_collided_device launch
        self.fused_reset = not self.no_reset and not args.separate_reset

    def set_gather(self, on, overlap=False, f32=False, root=None):
        """the RCCL observation gather after every step: off / in-stream / overlapped with the next step; f32: the scans
        cross the links as float32; root: only that rank receives (f110_comm_gather_obs).
        The communicator is created once (rank 0 makes the id, the control plane broadcasts it)."""
        sim, rdv = self.sim, self.rdv
        if self.d_all is not None:
            sim.sync()
            sim.comm_set_overlap(False)
        self.d_all = None
        if (f32, root) != (self.g_f32, self.g_root):   # other element type / other receivers: new receive buffers
            for d in self.d_alls + self.d_scals:
                d.free()
            self.d_alls, self.d_scals = [], []
        self.g_f32, self.g_root = bool(f32), root
        if not on:
            return
        if not self.comm_ready:
            from f1tenth_gym_amd import BatchSim as _B
            uid = _B.comm_unique_id() if rdv.rank == 0 else b"\0" * 128
            sim.comm_init(rdv.world, rdv.rank, rdv.broadcast_bytes(uid, 128))
            self.comm_ready = True
        want = 2 if overlap else 1
        recv = root is None or root == rdv.rank
        while len(self.d_alls) < want:   # receive buffers: scans [ranks][N][B] + scalars [ranks][7][N] (a non-root rank: one row, never written)
            self.d_alls.append(sim.device_array((rdv.world if recv else 1, self.N, self.beams), self.np.float32 if f32 else self.np.float64))
            self.d_scals.append(sim.device_array((rdv.world if recv else 1, 7, self.N)))
        self.n_recv = want
        self.d_all = self.d_alls[0]
        sim.comm_set_overlap(bool(overlap))

    def one(self, t, gather=True):
        sim = self.sim
        if self.d_zero is not None:
            sim.step_device(self.d_zero)
        elif self.planner is not None:
            self.planner.plan_device(sim, self.d_plan, 0.82461887897713965, 1.375 * 0.8)   # the example's look-ahead; 80 % of its speed gain
            sim.step_device(self.d_plan)
        else:
            sim.step_device(self.d_sets[t // 20])
        if self.d_all is not None and gather:
            if self.g_f32 or self.g_root is not None:
                sim.comm_gather_obs(self.d_alls[t % self.n_recv], self.d_scals[t % self.n_recv], f32=self.g_f32, root=self.g_root)
            else:
                sim.comm_all_gather_obs(self.d_alls[t % self.n_recv], self.d_scals[t % self.n_recv])
        if not self.no_reset and not self.fused_reset:
            sim.reset_collided_device(self.d_start, 0, self.d_count)

    def run(self, steps, warmup, mode="timed", preroll=None):
        """mode: timed (clean), profile (per-kernel HIP events), count (table lookups, counting kernels).
        Steps [0, preroll) and [preroll, preroll + warmup) are not timed; the pre-roll is this script's own
        (fixed), the warm-up the caller's."""
        np, sim, rdv = self.np, self.sim, self.rdv
        preroll = self.args.preroll if preroll is None else preroll
        first = preroll + warmup
        assert first + steps <= self.max_total
        sim.set_auto_reseat(None)
        sim.reset_device(self.d_start)
        if self.fused_reset:
            sim.set_auto_reseat(self.d_start, 0, self.d_count)
        for t in range(first):
            self.one(t, gather=t >= preroll)   # the pre-roll only ages the batch: no need to move 570 MB per rank per step for it
        sim.sync()
        self.d_count.upload(np.zeros(1, dtype=np.int32))
        if mode == "profile":
            sim.profile_kernels(True)
        if mode == "count":
            sim.scan_lookup_count(enable=True, read=True)
        rdv.barrier()
        sim.sync()
        t0 = time.perf_counter()
        sim.timer_begin()
        for t in range(first, first + steps):
            self.one(t)
        gpu_ms = sim.timer_end_ms()       # records + waits for the end event on the stream
        blocks = sim.step_groups()[2]     # env blocks the last timed step was submitted as (the library's choice unless --groups)
        sim.sync()
        mine = time.perf_counter() - t0   # this rank's own time (reported per rank); the job's time is max over ranks
        rdv.barrier()
        elapsed = time.perf_counter() - t0
        out = {"elapsed_s": elapsed, "rank_s": mine, "gpu_ms": gpu_ms, "n_reset": int(self.d_count.download()[0]), "steps": steps,
               "warmup": warmup, "preroll": preroll, "env_blocks": blocks}
        if mode == "profile":
            n, scan_ms, dyn_ms, fin_ms = sim.profile_read()
            out.update({"scan_ms_avg": scan_ms / max(n, 1), "dyn_ms_avg": dyn_ms / max(n, 1), "fin_ms_avg": fin_ms / max(n, 1), "n_prof": n})
            sim.profile_kernels(False)
        if mode == "count":
            out["lookups"] = sim.scan_lookup_count(enable=False)
        if self.d_all is not None and mode == "timed":
            # (with --separate-reset the re-seat follows the gather, so the state read back here is a later one)
            out["gather_ok"] = self.check_gather((first + steps - 1) % self.n_recv) if (self.fused_reset or self.no_reset) else None
        return out

    @staticmethod
    def digest(arr):
        """16 bytes that pin an array of doubles bit for bit: xor and wrapping sum of its 64-bit words"""
        import numpy as np
        w = np.ascontiguousarray(arr).view(np.uint64).reshape(-1)
        return struct.pack("<QQ", int(np.bitwise_xor.reduce(w)), int(np.sum(w, dtype=np.uint64)))

    def check_gather(self, slot):
        """after the last step: every rank's block in MY receive buffers must be that rank's own scans and
        scalar observation (each rank publishes digests of what it holds over the control plane)"""
        np, sim, rdv = self.np, self.sim, self.rdv
        o = sim.get("scans", "poses_x", "poses_y", "poses_theta", "linear_vels_x", "ang_vels_z", "collisions")
        scal = np.stack([o["poses_x"], o["poses_y"], o["poses_theta"], o["linear_vels_x"], np.zeros(self.N), o["ang_vels_z"], o["collisions"]])
        mine = o["scans"].astype(np.float32).astype(np.float64) if self.g_f32 else o["scans"]   # (digests are over 64-bit words)
        theirs = rdv.gather_bytes(self.digest(mine) + self.digest(scal))
        ok = True
        if self.g_root is None or self.g_root == rdv.rank:
            for r in range(rdv.world):   # one rank's block at a time: the receive buffer is world x 570 MB
                blk = self.d_alls[slot].download_part(r, 1)
                got = self.digest(blk.astype(np.float64) if self.g_f32 else blk) + self.digest(self.d_scals[slot].download_part(r, 1))
                ok = ok and got == theirs[r]
        return bool(ok)

    def close(self):
        for d in self.d_sets + [self.d_start, self.d_count] + [x for x in (self.d_plan, self.d_zero) if x is not None] + self.d_alls + self.d_scals:
            d.free()
        self.sim.close()


def scan_kernel_name(args, beams):
    aligned = args.layout == 3 and beams < 1498 and (-beams) % 64 * 100 <= 3 * beams
    if beams >= 1498:
        return "k_scan_dirs_agent" if args.layout == 3 else "k_scan_rays"
    return "k_scan_rays_agent" if aligned else "k_scan_rays"


def load_json(name):
    p = os.path.join(ROOT, "profiles", name)
    try:
        with open(p) as f:
            return json.load(f)
    except Exception:  # noqa: BLE001
        return None


def roofline_record(args, n_agents, beams, timed, prof, cnt, tiles=1):
    """SURVEY §8d.  timed / prof / cnt: the three passes over the same steps."""
    B = float(beams)
    steps = timed["steps"]
    agent_steps_per_s = n_agents * steps / timed["elapsed_s"]
    dedupe = beams >= 1498
    # lookups per ray the kernels performed over exactly the timed steps.  With more beams than
    # table directions the step marches each distinct direction once: the lookups are counted (and
    # the gathers priced) per distinct direction, so frac stays <= 1
    lbar = cnt["lookups"] / float(n_agents * B * cnt["steps"])
    b_stream = 216.0 + 8.0 * B
    b_alg = b_stream + 8.0 * B * lbar
    step_gbs = agent_steps_per_s * b_alg / 1e9
    # scalars first (the driver's parser keeps a bounded number of keys per object); the prose follows
    rec = {"bound": "hbm", "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_gbs / HBM_PEAK_GBS,
           "traffic": None, "kernel": None, "kernel_ms_avg": None, "kernel_frac": None, "issue_floor_frac": None, "hbm_measured_frac": None,
           "lookups_per_ray": lbar, "alg_bytes_per_agent_step": b_alg, "b_stream_bytes_per_agent_step": b_stream,
           "b_stream_frac": agent_steps_per_s * b_stream / 1e9 / HBM_PEAK_GBS,
           "binds": "ta-issue (L2-resident gathers): the table gathers hit L2 (TCC hit 96 %), the CUs' texture address / data path is 85-92 % "
                    "busy; `frac` prices the algorithmic bytes against the HBM peak as SURVEY 8d defines it, `issue_floor_frac` is the "
                    "fraction of the roofline that actually binds",
           "definition": "SURVEY 8d: agent-steps/s x B_alg / 8 TB/s over the timed steps; B_alg = 216 + 8*B + 8*B*L-bar bytes per agent-step",
           "lookups_counted": "on the device over the same %d steps after the same %d warm-up steps (replay with the counting kernels)%s"
                              % (cnt["steps"], cnt["warmup"], "; per distinct table direction (dedupe pass)" if dedupe else "")}
    key = "agents=%d,beams=%d,layout=%d" % (n_agents, beams, args.layout) + (",tiles=%d" % tiles if tiles > 1 else "")
    # evidence from profiles/ counts only when it was measured on THIS code: every entry carries the hash of the
    # kernel sources it was collected with (tools/summarize_prof.py), the library reports the hash it was built from
    from f1tenth_gym_amd import _ffi, build
    lib_csrc = (_ffi.lib().f110_build_info() or b"").decode().replace("csrc=", "")
    rec["csrc"] = lib_csrc
    if lib_csrc != build.src_hash():
        rec["csrc_note"] = "the loaded library was built from other sources (%s) than the tree holds (%s)" % (lib_csrc, build.src_hash())
    pmc = (load_json("pmc_scan.json") or {}).get(key)
    if pmc and pmc.get("csrc") != lib_csrc:
        rec["traffic"] = None
        rec["traffic_note"] = ("profiles/pmc_scan.json[%s] was collected on sources %s, this library is %s: not reported"
                               % (key, pmc.get("csrc", "unrecorded (%s)" % pmc.get("round", "?")), lib_csrc))
        pmc = None
    else:
        rec["traffic"] = pmc.get("hbm_bytes_per_launch") if pmc else None
        if pmc:
            rec["traffic_note"] = "rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) + WRITE_SIZE per launch, %s, sources %s (profiles/pmc_scan.json, %s)" % (
                pmc.get("window", "?"), pmc.get("csrc"), pmc.get("round", "?"))
        else:
            rec["traffic_note"] = "no PMC pass recorded for %s" % key
    if prof and "scan_ms_avg" in prof:
        scan_bytes = n_agents * (8.0 * B + 8.0 * B * lbar)   # range write + L-bar gathers of 8 B per ray
        k_ms = prof["scan_ms_avg"]
        k_gbs = scan_bytes / (k_ms * 1e-3) / 1e9
        rec.update({"kernel": scan_kernel_name(args, beams), "kernel_ms_avg": k_ms, "kernel_alg_bytes_per_launch": scan_bytes,
                    "kernel_achieved": k_gbs, "kernel_frac": k_gbs / HBM_PEAK_GBS,
                    "kernel_timing": "HIP events around every launch in a replay of the same steps (%d launches)" % prof["n_prof"]
                                     + ("; the timed steps ran as TWO env blocks on two streams (each kernel twice per step over half the envs, overlapping), the "
                                        "replay runs one block: the kernel_* entries describe one launch over the whole batch" if timed.get("env_blocks", 1) > 1 else ""),
                    "env_blocks_per_step": timed.get("env_blocks"),
                    "integrate_collide_ms_avg": prof["dyn_ms_avg"], "finalize_ms_avg": prof["fin_ms_avg"]})
        if pmc:
            rec["hbm_measured_frac"] = pmc["hbm_bytes_per_launch"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        fl = None
        for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_issue_floor.json")), reverse=True):
            cand = load_json(name)
            if cand and cand.get("csrc") == lib_csrc and key in cand.get("vmem_instr_per_launch", {}):
                fl = cand
                break
        if fl:
            vm = fl["vmem_instr_per_launch"][key]
            floor_ms = vm * fl["gather_cycles_per_wave_instr"] / fl["cus"] / (fl["clock_mhz"] * 1e3)
            rec["issue_floor"] = {"what": "the binding unit: wave-level vector-memory instructions x the cheapest a 64-lane gather can issue "
                                          "on a gfx950 CU (tools/debug/ta_bench.hip; measured per round: profiles/%s_ta_bench.txt)" % name.split("_")[0],
                                  "vmem_wave_instr_per_launch": vm, "cycles_per_instr": fl["gather_cycles_per_wave_instr"],
                                  "floor_ms": floor_ms, "kernel_ms": k_ms, "frac": floor_ms / k_ms, "csrc": fl.get("csrc"), "window": fl.get("window")}
            # what binds the kernel is the CUs' texture-address / data path issuing L2-resident gathers, not HBM: the
            # honest fraction-of-roofline of the dominant kernel is this one
            rec["issue_floor_frac"] = floor_ms / k_ms
        else:
            rec["issue_floor_note"] = "no *_issue_floor.json in profiles/ was measured on sources %s" % lib_csrc
    return rec


def parity_gate(args, rdv):
    """first 64 envs x 200 steps of the bench inputs: HIP vs oracle (flags exact, floats <= 1e-5)"""
    import numpy as np
    from f1tenth_gym_amd.workload import load_map_image
    from oracle import orc
    from f1tenth_gym_amd import BatchSim
    A, E, T = args.agents_per_env, 64, 200
    img, res, origin = load_map_image("example_map")
    dt, _, _ = oracle_map_dt("example_map")
    no_noise = args.no_noise or args.noise == "off"
    noise = None if no_noise else np.random.default_rng(NOISE_SEED).normal(0., 0.01, size=(T + 2, args.beams))
    sim = BatchSim(num_envs=E, num_agents=A, num_beams=args.beams, device_id=rdv.local_rank, map_layout=args.layout,
                   scan_block=args.scan_block, scan_tasks_per_wave=args.scan_tasks, step_groups=args.groups)
    sim.set_map_image(img, res, origin)
    ref = orc.SimOracle(E, A, num_beams=args.beams)
    ref.set_map_dt(dt, res, origin)
    if noise is not None:
        ref.set_noise(noise)          # the oracle adds NumPy's rows ...
        if args.noise == "table":
            sim.set_noise_table(noise)
        else:
            sim.set_noise_rng(NOISE_SEED, 0.01)   # ... the device draws them itself
    poses = start_poses_for(shard_envs(E, 0), A)
    sim.reset(poses); ref.reset(poses)
    sets = action_sets((T + 19) // 20, E * A, seed=1000)
    flag_mismatch, es, er = 0, 0.0, 0.0
    threads = min(os.cpu_count() or 1, 16)
    for t in range(T):
        sim.step(sets[t // 20]); ref.step(sets[t // 20], threads)
        if not args.no_reset:
            mask = (ref.collisions.reshape(E, A)[:, 0] != 0).astype(np.uint8)
            sim.reset(poses, mask); ref.reset(poses, mask)
        if t % 8 == 7 or t == T - 1:
            o = sim.get("scans", "state", "collisions", "in_collision")
            flag_mismatch += int(np.sum(o["collisions"] != ref.collisions) + np.sum(o["in_collision"] != ref.in_collision))
            es = max(es, rel_err(o["state"], ref.state))
            er = max(er, rel_err(o["scans"], ref.scans))
    sim.close()
    return {"envs": E, "steps": T, "flag_mismatches": flag_mismatch, "max_rel_err_state": es, "max_rel_err_scan": er,
            "tolerance": "|hip - oracle| <= 1e-5*|oracle| + 1e-12 per element (north_star: 1e-5 relative); flags exact",
            "ok": bool(flag_mismatch == 0 and es < 1e-5 and er < 1e-5)}


def cpu_baseline(args, seconds):
    """the CPU oracle (C port of the reference) on a bounded sample of the same workload, in the CPU's best shape:
    compiled for this machine (-O3 -march=native, the strict float64 flags kept: oracle/orc.py native_lib), every env
    walked through all its steps by one thread with the re-seats inside the C loop (orc_sim_rollout: no barrier per
    step, no Python between steps), OpenMP dynamic over envs on every CPU this rank may run on.  Beside it: the same
    code on ONE thread with the same envs (-> the 1 -> N scaling factor), and cpu_1t: the reference's own shape —
    1 env x 2 agents, one thread (BASELINE configs[0], SURVEY 8d "Config 1")."""
    import numpy as np
    from oracle import orc
    A = args.agents_per_env
    dt, res, origin = oracle_map_dt("example_map")
    no_noise = args.no_noise or args.noise == "off"
    per_set = 20

    def leg(E, threads, budget, native=True, max_steps=400):
        ref = orc.SimOracle(E, A, num_beams=args.beams, native=native)
        ref.set_map_dt(dt, res, origin)
        if not no_noise:
            ref.set_noise(np.random.default_rng(NOISE_SEED).normal(0., 0.01, size=(max_steps + 2, args.beams)))
        poses = start_poses_for(shard_envs(E, 0), A)
        ref.reset(poses)
        sets = np.stack(action_sets((max_steps + per_set - 1) // per_set, E * A, seed=1000))
        ref.rollout(sets, 2, per_set, poses, not args.no_reset, threads)    # first touch of the arrays, thread pool
        ref.reset(poses)
        look0 = ref.lookups
        steps = n_reset = 0
        t0 = time.perf_counter()
        while steps < max_steps - per_set:          # whole action sets until the budget is spent
            k = steps // per_set
            n_reset += ref.rollout(sets[k:k + 1], per_set, per_set, poses, not args.no_reset, threads)
            steps += per_set
            if time.perf_counter() - t0 >= budget:
                break
        el = time.perf_counter() - t0
        return E * A * steps / el, steps, el, (ref.lookups - look0) / float(steps * E * A * args.beams), n_reset

    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    # a container's CPU-time quota (cgroup cpu.max / cfs_quota): more runnable threads than that only get throttled
    # (measured on the GPU box, quota 16 of 256 CPUs: 154 k agent-steps/s at 16 threads, 104-135 k at 128)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
        except (OSError, ValueError):
            pass
    threads = max(1, len(cpus) if quota is None else min(len(cpus), int(quota + 0.5)))
    phys = None
    try:   # physical cores among the CPUs this rank may use (SMT siblings share one)
        seen = set()
        for c in cpus:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                seen.add(f.read().strip())
        phys = len(seen)
    except OSError:
        pass
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), None)
    except OSError:
        pass
    E = max(256, 16 * threads)
    # two builds of the same source: for this machine (-O3 -march=native) and the portable checker library (-O2); the
    # faster one is the baseline (on AVX-512 Xeons the -O2 build has been the faster one)
    half = max(2.0, seconds / 2.0)
    legs = {"native": leg(E, threads, half, native=True), "portable": leg(E, threads, half, native=False)}
    best = max(legs, key=lambda k: legs[k][0])
    v, steps, el, lbar, n_reset = legs[best]
    flags = " ".join(orc.NATIVE_CFLAGS[:2]) if best == "native" else "-O2"
    vs, steps_s, el_s, _, _ = leg(E, 1, min(seconds, 6.0), native=(best == "native"), max_steps=60)   # the same envs, same build, one thread
    v1, steps1, el1, _, _ = leg(1, 1, min(seconds, 5.0), native=(best == "native"), max_steps=20000)
    return {"value": v, "unit": "agent-steps/s", "cores": threads, "cpus_in_affinity": len(cpus), "physical_cores_in_affinity": phys,
            "cgroup_cpu_quota": quota, "cpu_model": model, "kind": "port",
            "sample": "%d envs x %d agents x %d steps of the bench workload, %d re-seats (oracle/f110_oracle.c orc_sim_rollout, gcc %s "
                      "-ffp-contract=off -fno-builtin, OpenMP dynamic over envs, every env through all its steps on one thread), %.1f s"
                      % (E, A, steps, n_reset, flags, el),
            "lookups_per_ray_on_sample": lbar,
            "builds": {k: {"value": r[0], "steps": r[1], "seconds": r[2]} for k, r in legs.items()}, "build_used": best,
            "scaling_1_to_n": {"threads": threads, "one_thread_value": vs, "factor": v / vs,
                               "sample_one_thread": "%d steps of the same %d envs, %.1f s" % (steps_s, E, el_s)},
            "cpu_1t": {"value": v1, "unit": "agent-steps/s", "cores": 1, "kind": "port",
                       "sample": "BASELINE configs[0] shape: 1 env x %d agents, %d steps, one thread, %.1f s (the reference runs this "
                                 "shape in numba on one core; numba cannot be installed here, so the C restatement stands in)" % (A, steps1, el1)}}


def dropin_rates(args):
    """The paths an RL loop actually calls, timed on the HIP library (wall clock around the Python calls):
    (i) BASELINE configs[0] as the drop-in runs it — F110Env(num_agents=2).step(action) on ONE env, host actions in,
    the reference's observation dict (scans included) out, episode logic on the host, exactly f110_env.py:263-304;
    (ii) F110VecEnv(E, device_logic=True, auto_reset=True): host actions in, `done` + lap arrays out, observations
    left in HBM — one f110_step_host call per step."""
    import numpy as np
    import f1tenth_gym_amd as amd
    from f1tenth_gym_amd.workload import PKG_MAPS as MAPS
    kw = dict(map=os.path.join(MAPS, "example_map"), map_ext=".png", num_agents=args.agents_per_env)
    out = {}

    def timed(fn, n, warm):
        for _ in range(warm):
            fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t0) / n
    A = args.agents_per_env
    env = amd.F110Env(**kw)
    env.reset(start_poses_for(shard_envs(1, 0), A))
    act = np.array([[0.05, 3.0], [-0.05, 2.5]] * A)[:A]
    for _ in range(200):
        env.step(act)
    env.sim.batch.step_host_stats()
    dt = timed(lambda: env.step(act), 3000, 0)
    _, enq, wait = env.sim.batch.step_host_stats()
    launches = env.sim.batch.step_launches()
    env.sim.batch.close()
    out["f110env_1env"] = {"workload": "F110Env(num_agents=%d).step(action): 1 env, host actions, obs dict with scans (BASELINE configs[0] through the HIP path), "
                                       "the first 3200 steps of a fresh env" % A,
                           "us_per_step": 1e6 * dt, "env_steps_per_s": 1.0 / dt, "value": A / dt, "unit": "agent-steps/s",
                           "in_f110_step_host_us": {"enqueue": enq, "wait": wait}, "python_around_the_call_us": 1e6 * dt - enq - wait,
                           "kernel_launches_per_step": {1: "1 (k_step_tiny)", 0: "per-kernel form"}.get(launches, launches)}
    # the reference's example loop (examples/waypoint_follow.py:272-285): ONE car
    env = amd.F110Env(**dict(kw, num_agents=1))
    env.reset(start_poses_for(shard_envs(1, 0), 1))
    dt1 = timed(lambda: env.step(np.array([[0.05, 3.0]])), 3000, 200)
    env.sim.batch.close()
    out["f110env_1env_1car"] = {"workload": "F110Env(num_agents=1).step(action): the reference's example loop shape", "us_per_step": 1e6 * dt1, "value": 1.0 / dt1,
                                "unit": "agent-steps/s"}
    for E in (2048, 32768):
        poses = start_poses_for(shard_envs(E, 0), A).reshape(E, A, 3)
        act = np.stack([a.reshape(E, A, 2) for a in action_sets(1, E * A, seed=1000)])[0]
        rec = {}
        for name, ekw, inplace in (("step_actions", {}, False), ("inplace_actions_lean", {"episode_fields": ()}, True),
                                   ("inplace_actions_lean_spin", {"episode_fields": (), "spin_wait": True}, True)):
            venv = amd.F110VecEnv(E, auto_reset=True, device_logic=True, obs_fields=(), **ekw, **kw)
            venv.reset(poses)
            n = 600 if E <= 4096 else 150
            if inplace:
                venv.action_buffer[...] = act
                dt = timed(lambda: venv.step(None), n, 50)
            else:
                dt = timed(lambda: venv.step(act), n, 50)
            _, enq, wait = venv.sim.batch.step_host_stats()
            venv.sim.batch.close()
            rec[name] = {"ms_per_step": 1e3 * dt, "value": E * A / dt, "host_enqueue_us": enq, "host_wait_us": wait}
        out["vecenv_%d" % E] = dict(rec, workload="F110VecEnv(%d envs x %d, device_logic=True, auto_reset=True, obs_fields=()): step_actions = step(ndarray) with the "
                                    "default episode fields; inplace_actions_lean = actions written into env.action_buffer, episode_fields=() (done only); "
                                    "_spin = completion word polled instead of hipStreamSynchronize" % (E, A), unit="agent-steps/s")
    # (iii) the scans CONSUMED where they are: a device-resident loop (examples/rl_loop_device.py) — the built-in reactive policy
    # reads all E*A*1080 ranges of the step just taken and writes the action buffer, the step with the episode logic follows,
    # finished envs are re-seated — nothing crosses PCIe, no host synchronisation inside the timed region
    for E in (2048, 32768):
        b = amd.BatchSim(num_envs=E, num_agents=A)
        b.set_map(os.path.join(MAPS, "example_map.yaml"), ".png")
        b.set_noise_rng(12345, 0.01)
        b.episode_init(0)
        b.episode_reset(start_poses_for(shard_envs(E, 0), A))
        d_act = b.device_array((E * A, 2)); d_act.upload(np.zeros((E * A, 2)))
        rec = {}
        for name, fn in (("scan_policy", lambda: b.scan_policy_device(d_act)), ("fixed_actions", lambda: None)):
            def one():
                fn(); b.episode_step_device(d_act); b.episode_reset_done_device()
            for _ in range(200):
                one()
            b.sync()
            n = 600 if E <= 4096 else 200
            t0 = time.perf_counter()
            for _ in range(n):
                one()
            b.sync()
            dt = (time.perf_counter() - t0) / n
            rec[name] = {"ms_per_step": 1e3 * dt, "value": E * A / dt}
        b.close()
        out["device_consumer_%d" % E] = dict(rec, unit="agent-steps/s", workload="%d envs x %d: scan_policy = f110_scan_policy_device (reads every scan in HBM, writes the "
                                             "action buffer) + f110_episode_step_device + f110_episode_reset_done_device per step, one sync at the end; fixed_actions = the "
                                             "same loop without the policy (what reading %d MB of scans per step costs)" % (E, A, E * A * 1080 * 8 >> 20))
    return out


def stub_run(args, rdv, steps, leg="headline"):
    """tests (no GPU): every rank 'steps' by sleeping; rank r pretends to be slower by r ms, the gather legs by 50 / 25 %"""
    if os.environ.get("F110_BENCH_STUB_FAIL") == leg and rdv.rank == rdv.world - 1:
        raise RuntimeError("stub: the %s leg fails on rank %d" % (leg, rdv.rank))   # tests: a leg that dies on one rank
    rdv.barrier()
    t0 = time.perf_counter()
    time.sleep(0.001 * steps * {"headline": 1.0, "gather": 1.5, "gather_overlap": 1.25}.get(leg, 1.2) + 0.001 * rdv.rank)
    mine = time.perf_counter() - t0
    rdv.barrier()
    return {"elapsed_s": time.perf_counter() - t0, "rank_s": mine, "n_reset": rdv.rank + 1, "steps": steps, "warmup": args.warmup,
            "preroll": args.preroll, "gather_ok": None if leg == "headline" else True}


def guarded(fn, timeout_s):
    """fn() in a worker thread (ctypes calls release the GIL) -> (result, None) | (None, what went wrong / 'timed out')"""
    box = {}

    def run():
        try:
            box["v"] = fn()
        except BaseException as ex:  # noqa: BLE001
            box["e"] = "%s: %s" % (type(ex).__name__, ex)
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        return None, "no answer within %d s" % timeout_s
    return box.get("v"), box.get("e")


def leg_record(rdv, total_agents, t):
    """one timed leg -> the job's numbers: the time is the MAX over ranks (barrier to barrier), per-rank own times beside it"""
    elapsed = rdv.max(t["elapsed_s"])
    per_rank = [1e3 * x / t["steps"] for x in rdv.gather(t["rank_s"])]
    oks = rdv.gather(-1.0 if t.get("gather_ok") is None else float(bool(t["gather_ok"])))
    return {"value": total_agents * t["steps"] / elapsed, "ms_per_step": 1e3 * elapsed / t["steps"],
            "per_rank_ms_per_step": per_rank, "per_rank_ms_per_step_min": min(per_rank), "per_rank_ms_per_step_max": max(per_rank),
            "env_resets_in_timed_region": int(rdv.sum(t["n_reset"])), "env_blocks": t.get("env_blocks"),
            "gather_ok": None if all(o < 0 for o in oks) else bool(all(o != 0.0 for o in oks))}


def main(argv=None, ident=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.no_noise:
        args.noise = "off"
    if ident is None and args.gpus > 1 and args.ranks_in_process:
        return thread_ranks(args, argv)
    if ident is None and args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args, argv)
    rdv = Rendezvous(ident)
    if rdv.world != args.gpus:
        if rdv.rank == 0:
            print("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, rdv.world), file=sys.stderr)
        return 2
    n_gpus = rdv.world
    numa = {"pci": None, "numa_node": -1, "cpus_bound": None, "note": "not attempted"}
    if not args.stub:
        import __graft_entry__
        if rdv.rank == 0 and not rdv.in_process:   # (thread_ranks has built already)
            __graft_entry__.build()
        rdv.barrier()
        from f1tenth_gym_amd import _ffi
        if _ffi.device_count() <= rdv.local_rank:
            raise SystemExit("bench.py: rank %d needs HIP device %d, %d visible" % (rdv.rank, rdv.local_rank, _ffi.device_count()))
        if rdv.in_process:
            numa["note"] = "ranks are threads of one process: no per-rank pinning"
        elif not args.no_numa:
            from f1tenth_gym_amd import numa as _numa
            numa = _numa.bind_to_device(rdv.local_rank)   # before the first launch: the HIP runtime's threads inherit it

    extras = n_gpus == 1 and not args.only_headline and not args.stub
    total_needed = args.preroll + args.warmup + args.steps
    if extras and args.steady_steps > 0:
        total_needed = max(total_needed, args.preroll + args.steady_warmup + args.steady_steps)
    total_agents = args.agents * n_gpus
    # the legs of ONE invocation: the headline (no data-path collective unless --gather asks for it), then — on
    # more than one GPU — the same steps with the RCCL observation gather in-stream and overlapped (SURVEY 8e:
    # "report scaling both with and without it")
    gather_legs = (n_gpus > 1 or args.gather_legs) and not args.no_gather_legs and not args.gather and not args.only_headline
    wl = None
    rccl_ranks = None
    if args.stub:
        timed = stub_run(args, rdv, args.steps)
        head = leg_record(rdv, total_agents, timed)
    else:
        wl = Workload(args, rdv, args.agents, total_needed)
        timed = wl.run(args.steps, args.warmup, "timed")
        head = leg_record(rdv, total_agents, timed)
        if args.gather:
            rccl_ranks = wl.sim.comm_info()[0]
    elapsed = head["ms_per_step"] * args.steps / 1e3
    value = head["value"]
    n_reset = head["env_resets_in_timed_region"]
    if numa.get("note"):
        numa["note"] = str(numa["note"])[:120]
    numa_all = [json.loads(b.rstrip(b"\0").decode()) for b in rdv.gather_bytes(json.dumps(numa).encode().ljust(512, b"\0"))]

    line = {
        "metric": "agent-steps/s (1080-beam scan + ST dynamics)", "value": value, "unit": "agent-steps/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("%d agents per GPU (%d envs x %d), example_map%s @0.0625 m, %d-beam lidar, "
                                "ST dynamics RK4 dt=0.01, scan noise %s, in-place resets %s"
                                % (args.agents, args.agents // args.agents_per_env, args.agents_per_env,
                                   " 1600x1600" if args.map_tiles == 1 else " tiled %dx%d = %dx%d cells" % (args.map_tiles, args.map_tiles, 1600 * args.map_tiles, 1600 * args.map_tiles),
                                   args.beams,
                                   {"rng": "on (drawn on the device: NumPy's seed-12345 PCG64/ziggurat stream)", "table": "on (NumPy rows uploaded)", "off": "off"}[args.noise],
                                   "off" if args.no_reset else "on"))
                               + (" (BASELINE configs[2])" if args.agents == 65536 and args.beams == 1080 else ""),
                   "policy": {"random": "pre-drawn random actions, device resident",
                              "pure_pursuit": "reference pure-pursuit planner evaluated on the device every step (closed loop, planner time included)",
                              "parked": "zero actions: every car stays on its start pose"}[args.policy],
                   "agents_per_gpu": args.agents, "agents_total": total_agents, "beams": args.beams,
                   "map_layout": {0: "rowmajor_f64", 3: "padded_rowmajor_f64"}[args.layout],
                   "scan_block": args.scan_block, "scan_tasks_per_wave": args.scan_tasks, "step_groups": args.groups,
                   "env_blocks_per_step": head.get("env_blocks"),   # 1 = every kernel once per step over the whole batch; 2 = two halves of the envs on two streams (include/f110.h step_groups)
                   "parallelism": "env-sharded x%d, %s" % (n_gpus, ("RCCL all-gather of the observation (scans + 7 scalars per agent) after every step" + (" (overlapped with the next step, double-buffered)" if args.gather_overlap else "")) if args.gather
                                                           else "no data-path collective"),
                   "preroll_steps": args.preroll,
                   "timed_window": "steps [%d, %d) after the reset: %d un-timed pre-roll steps (fixed by this script, so the timed steps run in the "
                                   "steady regime with in-place resets), then the caller's %d warm-up steps" % (args.preroll + args.warmup, args.preroll + args.warmup + args.steps, args.preroll, args.warmup),
                   "env_resets_in_timed_region": int(n_reset)},
        # one process per GPU: each rank's own time for the timed steps (the job's time above is the max, barrier to barrier)
        "multi_gpu": {"per_rank_ms_per_step": head["per_rank_ms_per_step"], "per_rank_ms_per_step_min": head["per_rank_ms_per_step_min"],
                      "per_rank_ms_per_step_max": head["per_rank_ms_per_step_max"], "per_gpu_value": value / n_gpus,
                      "rccl_ranks": rccl_ranks, "numa": numa_all,
                      "ranks_are": "threads of one process, one handle each (--ranks-in-process)" if rdv.in_process else "one process per GPU",
                      "legs": "headline = no collective on the step path" + ("; gather = f110_comm_all_gather_obs after every step on the step's stream; "
                              "gather_overlap = the same on a stream of its own beside the next step (double-buffered observation); gather_f32 = the scans "
                              "cross the links as float32 (f110_comm_gather_obs); gather_root = only rank 0 receives (grouped ncclSend / ncclRecv)" if gather_legs else "")},
    }
    if args.gather:
        line["config"]["gather_ok"] = head["gather_ok"]

    if wl is not None and rdv.rank == 0 and not args.only_headline:
        # the same steps twice more on rank 0: per-kernel HIP events, then the counting kernels
        solo = Rendezvous.__new__(Rendezvous)
        solo.world, solo.rank, solo.local_rank, solo.peers, solo.sock = 1, 0, rdv.local_rank, [], None
        wl.rdv = solo
        if wl.d_all is not None and n_gpus > 1:
            wl.set_gather(False)   # rank 0 replays alone: no collective may be enqueued
        prof = None if args.no_profile_events else wl.run(args.steps, args.warmup, "profile")
        cnt = wl.run(args.steps, args.warmup, "count")
        line["roofline"] = roofline_record(args, args.agents, args.beams, dict(timed, elapsed_s=elapsed), prof, cnt, args.map_tiles)
        if extras and args.steady_steps > 0:
            st = wl.run(args.steady_steps, args.steady_warmup, "timed")
            sp = wl.run(args.steady_steps, args.steady_warmup, "profile")
            sc = wl.run(args.steady_steps, args.steady_warmup, "count")
            line["steady_state"] = {"value": args.agents * args.steady_steps / st["elapsed_s"], "unit": "agent-steps/s",
                                    "steps": args.steady_steps, "warmup": args.steady_warmup,
                                    "definition": "SURVEY 8d: >= 1000 timed steps after >= 100 warm-up steps — here %d pre-roll + %d warm-up steps, the same "
                                                  "regime the headline's short window samples (measured: 20 timed steps after 0 / 100 / 300 / 500 / 1000 / 2000 "
                                                  "un-timed ones run at 103.0 / 102.9 / 96.0 / 95.0 / 95.4 / 95.0 M agent-steps/s: the batch needs ~300 steps "
                                                  "to reach it)" % (args.preroll, args.steady_warmup),
                                    "headline_over_steady": value / (args.agents * args.steady_steps / st["elapsed_s"]),
                                    "ms_per_step": 1e3 * st["elapsed_s"] / args.steady_steps, "env_resets_in_timed_region": st["n_reset"],
                                    "roofline": roofline_record(args, args.agents, args.beams, st, sp, sc, args.map_tiles)}
        wl.rdv = rdv

    if gather_legs:
        # The gather legs come LAST and under a watchdog: the headline (and rank 0's replays of it) are already in
        # `line`, so whatever the communicator does on this node — refuses to initialise, or never returns — the
        # run still ends with its one line, `multi_gpu.gather_error` saying what happened.
        rdv.barrier()     # (the other ranks have been waiting here for rank 0's replays)
        err = None
        t_legs = time.perf_counter()

        def gather_pair(work, agents_per_rank, out):
            """the same steps with the observation gather in the step's stream, then overlapped; -> rccl_ranks"""
            for name, overlap, f32, root in (("gather", False, False, None), ("gather_overlap", True, False, None),
                                             ("gather_f32", False, True, None), ("gather_f32_overlap", True, True, None),
                                             ("gather_root", False, False, 0), ("gather_root_f32_overlap", True, True, 0)):
                # one decision for all ranks (rank 0's clock): a leg that would start after the budget is skipped and listed
                late = rdv.max(1.0 if (rdv.rank == 0 and time.perf_counter() - t_legs > args.gather_budget) else 0.0) > 0.0
                if late:
                    out.setdefault("legs_skipped", []).append(name)
                    continue
                ranks_now = None
                if args.stub:
                    res = stub_run(args, rdv, args.steps, name)
                else:
                    res, e = guarded(lambda: (work.set_gather(True, overlap, f32, root), work.run(args.steps, args.warmup, "timed"))[1], args.gather_timeout)
                    if e:
                        raise RuntimeError("%s leg: %s" % (name, e))
                    ranks_now = work.sim.comm_info()[0]     # ncclCommCount of the communicator this leg's collectives ran on
                rec = leg_record(rdv, agents_per_rank * n_gpus, res)
                esz = 4 if f32 else 8
                if not args.stub:   # this rank's device while the leg's receive buffers are allocated (all ranks' memory when they share one device)
                    free_b, total_b = work.sim.device_mem_info()
                    rec["device_mem_used_gb_max"] = rdv.max((total_b - free_b) / 1e9)
                out[name] = dict(rec, per_gpu_value=rec["value"] / n_gpus, rccl_ranks=ranks_now,
                                 bytes_received_per_step={"root" if root is not None else "every_rank": agents_per_rank * (esz * args.beams + 56) * n_gpus},
                                 bytes_sent_per_rank_per_step=agents_per_rank * (esz * args.beams + 56))
            if args.stub:
                return None
            n = work.sim.comm_info()[0]
            work.set_gather(False)
            return n
        try:
            rccl_ranks = gather_pair(wl, args.agents, line["multi_gpu"])
            line["multi_gpu"]["rccl_ranks"] = rccl_ranks
            c3_agents = 262144 // n_gpus
            if (n_gpus == 8 or args.config3_legs) and c3_agents != args.agents and c3_agents % args.agents_per_env == 0:
                # BASELINE configs[3] to the letter: 262 144 agents over the node (32 768 per GPU at 8), without and
                # with the gather — the legs above keep the scaling curve's 65 536 agents per GPU
                c3 = {"workload": "262144 agents sharded x%d = %d per GPU, RCCL gather of the observation (BASELINE configs[3])" % (n_gpus, c3_agents)}
                if wl is not None:
                    wl.close()
                    wl = None
                w3 = None if args.stub else Workload(args, rdv, c3_agents, args.preroll + args.warmup + args.steps)
                if args.stub:
                    res = stub_run(args, rdv, args.steps)
                else:
                    res, e = guarded(lambda: w3.run(args.steps, args.warmup, "timed"), args.gather_timeout)
                    if e:
                        raise RuntimeError("configs[3] leg: %s" % e)
                rec = leg_record(rdv, c3_agents * n_gpus, res)
                c3["no_gather"] = dict(rec, per_gpu_value=rec["value"] / n_gpus)
                gather_pair(w3, c3_agents, c3)
                line["multi_gpu"]["configs3"] = c3
                if w3 is not None:
                    w3.close()
        except BaseException as ex:  # noqa: BLE001 — incl. "peer went away" when another rank gave up
            err = "%s: %s" % (type(ex).__name__, ex)
        if err:
            line["multi_gpu"]["gather_error"] = err[:400]
            if rdv.rank == 0:
                print(json.dumps(line))
                sys.stdout.flush()
            os._exit(0)   # a collective may be stuck on this handle's streams: no orderly teardown is possible

    if wl is not None:
        wl.close()

    if extras and rdv.rank == 0:
        def other(n_agents, steps, warmup, **kw):
            w2 = Workload(args, rdv, n_agents, args.preroll + steps + warmup, **kw)
            t = w2.run(steps, warmup, "timed")
            p = w2.run(steps, warmup, "profile")
            c = w2.run(steps, warmup, "count")
            w2.close()
            return t, p, c
        if args.secondary and args.secondary != args.agents:
            k2 = max(args.steps, 300)
            t, p, c = other(args.secondary, k2, max(args.warmup, 30))
            sec_roof = roofline_record(args, args.secondary, args.beams, t, p, c)
            flat = line["config"]   # flat scalars: the driver's parser keeps scalars and drops nested objects
            flat["configs1_agents"], flat["configs1_value"] = args.secondary, args.secondary * k2 / t["elapsed_s"]
            flat["configs1_frac"], flat["configs1_issue_floor_frac"] = sec_roof["frac"], sec_roof.get("issue_floor_frac")
            line["config"]["secondary"] = {"workload": "%d agents (BASELINE configs[1])" % args.secondary,
                                           "value": args.secondary * k2 / t["elapsed_s"], "ms_per_step": 1e3 * t["elapsed_s"] / k2,
                                           "steps": k2, "env_resets_in_timed_region": t["n_reset"],
                                           "roofline": sec_roof}
        if not args.no_config5 and args.beams == 1080 and args.map_tiles == 1 and args.agents == 65536:
            k5 = 100
            t, p, c = other(65536, k5, 20, beams=4096, map_tiles=2)
            c5_roof = roofline_record(args, 65536, 4096, t, p, c, tiles=2)
            flat = line["config"]
            flat["configs4_value"], flat["configs4_frac"] = 65536 * k5 / t["elapsed_s"], c5_roof["frac"]
            flat["configs4_issue_floor_frac"] = c5_roof.get("issue_floor_frac")
            line["config"]["config5"] = {"workload": "65536 agents, 4096 beams, example_map tiled 2x2 = 3200x3200 cells (BASELINE configs[4])",
                                         "value": 65536 * k5 / t["elapsed_s"], "ms_per_step": 1e3 * t["elapsed_s"] / k5, "steps": k5,
                                         "env_resets_in_timed_region": t["n_reset"],
                                         "roofline": c5_roof}
        if args.fixed_pose_steps > 0 and args.policy == "random":
            w3 = Workload(args, rdv, args.agents, args.fixed_pose_steps + 10, policy="parked", no_reset=True)
            r3 = w3.run(args.fixed_pose_steps, 10, "timed", preroll=0)
            w3.close()
            line["config"]["fixed_pose_variant"] = {"workload": "same agents parked on their start poses (speed 0, no resets)",
                                                    "value": args.agents * args.fixed_pose_steps / r3["elapsed_s"],
                                                    "ms_per_step": 1e3 * r3["elapsed_s"] / args.fixed_pose_steps}
        if not args.no_dropin and args.beams == 1080 and args.map_tiles == 1:
            line["config"]["dropin"] = dropin_rates(args)
            line["config"]["configs0_f110env_us_per_step"] = line["config"]["dropin"]["f110env_1env"]["us_per_step"]
            line["config"]["configs0_f110env_1car_us_per_step"] = line["config"]["dropin"]["f110env_1env_1car"]["us_per_step"]
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
            line["parity_gate"] = parity_gate(args, rdv)
    if rdv.rank == 0:
        print(json.dumps(line))
        sys.stdout.flush()
    rdv.barrier()
    rdv.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
