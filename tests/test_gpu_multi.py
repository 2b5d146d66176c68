"""-m gpu tests of the N-rank path that switch themselves on when the box has the devices:

* 2 ranks on 2 devices (skipped cleanly on a 1-GPU box): every rank verifies the OTHER rank's gathered
  observation (scans + 7 scalars per agent, f110_comm_all_gather_obs) bit for bit against a local
  re-simulation of that shard — in the step's stream and overlapped with the next step;
* the same two ranks sharing ONE device, which exercises a real 2-rank RCCL communicator on a 1-GPU box
  when RCCL accepts it (it may refuse two ranks on one device: then the test reports that and skips);
* bench.py at --gpus 2 on 2 devices: one invocation, the three legs (no gather / gather / overlapped),
  per-rank times, the communicator size as RCCL reports it.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _devices():
    from f1tenth_gym_amd import _ffi
    return _ffi.device_count()


def _partition_note():
    """why a 1-GPU box shows one device: the compute partition mode (SPX = one logical device per MI355X; in CPX mode the same GPU
    shows eight, and the 2- / 4-device tests below then run real multi-rank RCCL on partitions of one package)"""
    try:
        out = subprocess.run(["rocm-smi", "--showcomputepartition"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=30).stdout
        modes = sorted({w for line in out.splitlines() if "artition" in line for w in line.replace(":", " ").split() if w in ("SPX", "DPX", "TPX", "QPX", "CPX")})
        return "compute partition mode: %s" % (", ".join(modes) if modes else "not reported")
    except Exception as ex:  # noqa: BLE001
        return "compute partition mode unknown (%s)" % type(ex).__name__


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


def _launch(world, cmd, extra_env=None, timeout=900):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    try:
        outs = [p.communicate(timeout=timeout) for p in procs]
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        for p in procs:
            p.communicate()
        return procs, None
    return procs, outs


def _results(procs, outs):
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    res = []
    for so, _ in outs:
        lines = [l for l in so.splitlines() if l.startswith("RESULT ")]
        assert len(lines) == 1, so[-500:]
        res.append(json.loads(lines[0][7:]))
    return sorted(res, key=lambda r: r["rank"])


def _check_gather_results(res, world):
    for r in res:
        assert r["world"] == world and r["rccl_ranks"] == world and r["rccl_rank"] == r["rank"]
        assert r["in_stream_mismatches"] == 0, r
        assert r["overlapped_mismatches"] == 0, r
        assert r["steps"] >= 40   # >= 20 steps in each form


def test_two_ranks_two_devices_gather_the_observation():
    if _devices() < 2:
        pytest.skip("needs 2 HIP devices (this box has %d; %s)" % (_devices(), _partition_note()))
    procs, outs = _launch(2, [sys.executable, os.path.join(ROOT, "tests", "dist_worker.py")])
    res = _results(procs, outs)
    assert [r["device"] for r in res] == [0, 1]
    _check_gather_results(res, 2)


def test_four_ranks_four_devices_gather_the_observation():
    if _devices() < 4:
        pytest.skip("needs 4 HIP devices (this box has %d; %s)" % (_devices(), _partition_note()))
    procs, outs = _launch(4, [sys.executable, os.path.join(ROOT, "tests", "dist_worker.py")], {"F110_DIST_ENVS": "32"})
    _check_gather_results(_results(procs, outs), 4)


def test_two_ranks_sharing_one_device_gather_the_observation():
    """a real world-size-2 RCCL communicator on a 1-GPU box, if RCCL lets two ranks share a device"""
    procs, outs = _launch(2, [sys.executable, os.path.join(ROOT, "tests", "dist_worker.py")], {"F110_BENCH_DEVICE": "0"}, timeout=300)
    if outs is None:
        pytest.skip("two RCCL ranks on one device did not finish in 300 s (RCCL does not support sharing a device)")
    res = _results(procs, outs)
    if any("rccl_refused" in r for r in res):
        pytest.skip("RCCL refuses two ranks on one device: %s" % [r.get("rccl_refused") for r in res][0])
    _check_gather_results(res, 2)


def test_bench_two_gpus_one_invocation_three_legs():
    if _devices() < 2:
        pytest.skip("needs 2 HIP devices (this box has %d; %s)" % (_devices(), _partition_note()))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "F110_BENCH_RDV")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--agents", "8192",
                          "--preroll", "60"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-1500:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    mg = d["multi_gpu"]
    assert d["n_gpus"] == 2 and d["config"]["agents_total"] == 16384 and mg["rccl_ranks"] == 2
    assert len(mg["per_rank_ms_per_step"]) == 2 and len(mg["numa"]) == 2
    for leg in ("gather", "gather_overlap"):
        assert mg[leg]["gather_ok"] is True and mg[leg]["value"] > 0
    assert d["config"]["env_resets_in_timed_region"] >= 0 and "roofline" in d


@pytest.mark.parametrize("world", [2, 4])
def test_bench_ranks_sharing_one_device_without_the_gather(world):
    """`python bench.py --gpus N` end to end on REAL hardware with every rank on device 0 (F110_BENCH_DEVICE): the launcher
    in bench.py spawns the ranks, the socket control plane runs between real processes, every rank steps its own shard on
    the GPU, NUMA binding per rank, the job's time is the max over ranks, rank 0 prints the one line.  No collective
    (RCCL refuses several ranks on one device; the gather's N > 1 paths are covered by tests/rccl_stub)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "F110_BENCH_RDV")}
    env["F110_BENCH_DEVICE"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "40", "--warmup", "5", "--agents", "4096",
                          "--preroll", "60", "--no-gather-legs"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-1500:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-800:]
    d = json.loads(lines[0])
    mg = d["multi_gpu"]
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["config"]["agents_total"] == 4096 * world
    assert len(mg["per_rank_ms_per_step"]) == world and len(mg["numa"]) == world and all(t > 0 for t in mg["per_rank_ms_per_step"])
    assert abs(d["value"] - 4096 * world * 40 / (d["ms_per_step"] * 40e-3)) < 1e-6 * d["value"]       # all ranks' agent-steps over the max time
    assert d["ms_per_step"] >= max(mg["per_rank_ms_per_step"]) * 0.999
    assert "gather" not in mg and mg.get("gather_error") is None and "roofline" in d
    assert d["config"]["env_resets_in_timed_region"] > 0
