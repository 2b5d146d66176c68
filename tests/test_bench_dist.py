"""CPU-only, world_size 2: bench.py's rendezvous (barrier, max / sum / broadcast over ranks — stdlib
sockets, no torch), the env sharding used for N > 1 GPUs (contiguous env blocks, disjoint start
poses, per-rank seeds), bench.main() itself at world size 2 with a stubbed step — launched as ranks
(the driver's torchrun form) and self-launched (`python bench.py --gpus 2`)."""
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
blob = rdv.broadcast_bytes(bytes(range(128)) if rdv.rank == 0 else b"", 128)
out = {"rank": rdv.rank, "world": rdv.world, "max": rdv.max(1.0 + rdv.rank), "sum": rdv.sum(10.0 * (rdv.rank + 1)),
       "blob_ok": blob == bytes(range(128)),
       "ids": ids.tolist(), "pose0": poses[0].tolist(), "act0": acts[0][0].tolist()}
rdv.barrier()
print("RESULT " + json.dumps(out)); sys.stdout.flush()
rdv.close()
'''


def _free_port():
    """a port nobody listens on, taken BELOW the kernel's ephemeral range (32768+): a port from bind(0) can be handed to another process's
    outgoing connection between this probe and the rank that binds it"""
    import random
    for _ in range(200):
        p = random.randint(20000, 29999)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", p))
        except OSError:
            continue
        finally:
            s.close()
        return p
    raise RuntimeError("no free port in 20000..29999")


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
    assert outs[0]["blob_ok"] and outs[1]["blob_ok"]       # the RCCL unique id travels this way
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


def _bench_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_main_as_two_launched_ranks_with_stub_step():
    """what the driver does for N > 1: N processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT in
    the environment; rank 0 prints the one JSON line, value = all ranks' agent-steps over the MAX time"""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "3",
                                       "--agents", "1000", "--stub"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
    line = _bench_line(outs[0][0])
    assert outs[1][0].strip() == ""                         # only rank 0 prints
    assert line["n_gpus"] == 2 and line["steps"] == 40 and line["warmup"] == 3 and line["scaling"] == "weak"
    assert line["config"]["agents_total"] == 2000 and line["config"]["env_resets_in_timed_region"] == 3   # sum over ranks (1 + 2)
    assert abs(line["value"] - 2000 * 40 / (line["ms_per_step"] * 40e-3)) < 1e-6 * line["value"]
    assert line["ms_per_step"] * 40e-3 >= 0.041              # the slower rank's time (stub: 40 ms + 1 ms x rank)
    # ONE invocation carries the three legs SURVEY 8e asks for, each with every rank's own time
    mg = line["multi_gpu"]
    assert len(mg["per_rank_ms_per_step"]) == 2 and mg["per_rank_ms_per_step"][1] > mg["per_rank_ms_per_step"][0]
    assert mg["per_rank_ms_per_step_min"] == min(mg["per_rank_ms_per_step"]) and mg["per_rank_ms_per_step_max"] <= line["ms_per_step"] * 1.0001
    assert abs(mg["per_gpu_value"] - line["value"] / 2) < 1e-9 * line["value"]
    for leg in ("gather", "gather_overlap"):
        assert mg[leg]["gather_ok"] is True and len(mg[leg]["per_rank_ms_per_step"]) == 2
        assert mg[leg]["value"] < line["value"]                # the stub's gather legs are slower by 50 / 25 %
        assert mg[leg]["bytes_received_per_step"] == {"every_rank": 8 * 1000 * (1080 + 7) * 2}
        assert mg[leg]["bytes_sent_per_rank_per_step"] == 8 * 1000 * (1080 + 7)
    # round 4: float32 transport and gather-to-root as legs of their own (SURVEY 8e's 142 MB / one receiver)
    assert mg["gather_f32"]["bytes_sent_per_rank_per_step"] == 1000 * (4 * 1080 + 56)
    assert mg["gather_root"]["bytes_received_per_step"] == {"root": 8 * 1000 * (1080 + 7) * 2}
    assert set(mg) >= {"gather_f32_overlap", "gather_root_f32_overlap"}
    assert mg["gather_overlap"]["value"] > mg["gather"]["value"]
    assert mg["rccl_ranks"] is None                          # no communicator in the stub
    assert len(mg["numa"]) == 2
    assert line["config"]["preroll_steps"] == 500 and "pre-roll" in line["config"]["timed_window"]


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it spawns the two ranks itself"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "F110_BENCH_RDV")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "2", "--agents", "512",
                          "--stub"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1000:]
    line = _bench_line(out.stdout)
    assert line["n_gpus"] == 2 and line["config"]["agents_total"] == 1024


def test_configs3_to_the_letter_legs():
    """BASELINE configs[3] — 262 144 agents over the node with the observation gather — runs inside the same
    invocation (automatically at --gpus 8, on request elsewhere): without the gather, with it, overlapped"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "F110_BENCH_RDV")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "2", "--agents", "512",
                          "--stub", "--config3-legs"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1000:]
    c3 = _bench_line(out.stdout)["multi_gpu"]["configs3"]
    assert "131072 per GPU" in c3["workload"]
    for leg in ("no_gather", "gather", "gather_overlap"):
        assert c3[leg]["value"] > 0 and len(c3[leg]["per_rank_ms_per_step"]) == 2
    assert c3["gather"]["bytes_received_per_step"] == {"every_rank": 8 * 131072 * 1087 * 2}
    assert c3["no_gather"]["value"] > c3["gather_overlap"]["value"] > c3["gather"]["value"]


def test_a_failing_gather_leg_cannot_take_the_headline_with_it():
    """the gather legs run last and under a watchdog: when one dies on some rank (here: rank 1 raises in the
    overlapped leg, rank 0 then finds its peer gone) every rank still exits 0 and rank 0 still prints the line —
    the headline, the leg that did complete, and `gather_error`"""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.guarded(lambda: 7, 5.0) == (7, None)
    assert bench.guarded(lambda: 1 / 0, 5.0)[1].startswith("ZeroDivisionError")
    import time
    assert "no answer within" in bench.guarded(lambda: time.sleep(3), 0.2)[1]
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   F110_BENCH_STUB_FAIL="gather_overlap")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "2",
                                       "--agents", "100", "--stub"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
    line = _bench_line(outs[0][0])
    mg = line["multi_gpu"]
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert "gather" in mg and "gather_overlap" not in mg and mg["gather_error"]


def test_bench_single_rank_has_no_gather_legs_unless_asked():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "F110_BENCH_RDV")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "1", "--agents", "64", "--stub"]
    out = subprocess.run(base, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert out.returncode == 0, out.stderr[-1000:]
    mg = _bench_line(out.stdout)["multi_gpu"]
    assert "gather" not in mg and len(mg["per_rank_ms_per_step"]) == 1
    out = subprocess.run(base + ["--gather-legs"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert "gather" in _bench_line(out.stdout)["multi_gpu"]


def test_rendezvous_gather_in_rank_order():
    sys.path.insert(0, ROOT)
    import bench
    env = {k: os.environ.pop(k) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK") if k in os.environ}
    try:
        r = bench.Rendezvous()
        assert r.gather(2.5) == [2.5] and r.gather_bytes(b"ab") == [b"ab"]
        r.close()
    finally:
        os.environ.update(env)


def test_numa_binding_follows_sysfs(tmp_path):
    """f1tenth_gym_amd.numa: PCI bus id -> sysfs numa_node -> that node's cpulist (cut to the allowed set)"""
    from f1tenth_gym_amd import numa
    assert numa.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text("64-127\n")
    bound = []
    rec = numa.bind_to_node_of("0000:c1:00.0", sysfs=str(tmp_path), setaffinity=bound.append, allowed=range(0, 96))
    assert rec["numa_node"] == 1 and rec["cpus_bound"] == 32 and bound == [list(range(64, 96))]
    (dev / "numa_node").write_text("-1\n")   # single-node boxes / VMs
    rec = numa.bind_to_node_of("0000:c1:00.0", sysfs=str(tmp_path), setaffinity=bound.append)
    assert rec["numa_node"] == -1 and rec["cpus_bound"] is None and len(bound) == 1
    rec = numa.bind_to_node_of("0000:ff:00.0", sysfs=str(tmp_path), setaffinity=bound.append)   # unknown device
    assert rec["cpus_bound"] is None and len(bound) == 1


def test_bench_refuses_a_mismatched_world():
    """--gpus 4 inside a 2-rank launch must not print a line labelled with either number"""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "5", "--stub"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode != 0 for p in procs)
    assert not any(o[0].strip() for o in outs)


def test_committed_bench_line_meets_the_contract():
    """the last bench line committed under profiles/ (a real run on an MI355X) carries every key the driver's contract names —
    a renamed or dropped key shows up here, on the CPU, when the profile is refreshed"""
    import glob, json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")))
    assert files
    d = json.loads(open(files[-1]).readline())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f64" and "synthetic" in d["data"]
    assert d["unit"] == "agent-steps/s" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["agents_total"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] <= 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    assert d["parity_gate"]["ok"] is True


def test_gather_legs_respect_their_time_budget():
    """VERDICT r4 #6a: the gather legs of one invocation share a time budget (--gather-budget, default 240 s, so that the default
    `--gpus 8` run stays under five minutes whatever the links do): with a budget the first legs use up, the rest are skipped —
    the same decision on every rank — and listed; the line is still printed"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "F110_BENCH_RDV")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "60", "--warmup", "2", "--agents", "512",
                          "--stub", "--gather-budget", "0.1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1000:]
    mg = _bench_line(out.stdout)["multi_gpu"]
    assert "gather" in mg and mg["gather"]["value"] > 0          # the first leg starts inside the budget (60 stub steps ~ 0.09 s + 0.03)
    assert mg.get("legs_skipped") and mg["legs_skipped"][-1] == "gather_root_f32_overlap" and "gather" not in mg["legs_skipped"]
    assert not (set(mg["legs_skipped"]) & set(k for k in mg if k.startswith("gather")))
    assert mg.get("gather_error") is None
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "2", "--agents", "512", "--stub"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    mg = _bench_line(out.stdout)["multi_gpu"]
    assert "legs_skipped" not in mg and mg["gather_root_f32_overlap"]["rccl_ranks"] is None     # (stub: no communicator; real runs: ncclCommCount per leg)
