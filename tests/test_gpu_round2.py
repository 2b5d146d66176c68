"""GPU parity tests added in round 2 (-m gpu): the device noise generator against NumPy's stream,
env-group stepping against single-block stepping, BASELINE configs[4] through the step, the big
shipped maps, the lookup counter behind bench.py's L-bar, the RCCL gather at world size 1, and the
randomised fuzzers with bounded seeds.  Everything goes through the C ABI; nothing reads
/root/reference."""
import hashlib
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

from _util import gold, load_map_image, oracle_map_dt, bench_start_poses, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NORTH_STAR = 1e-5


@pytest.fixture(scope="module")
def amd():
    import f1tenth_gym_amd
    from f1tenth_gym_amd import _ffi
    assert _ffi.device_count() >= 1, "no MI355X visible: the HIP path cannot run (no CPU fallback)"
    return f1tenth_gym_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import orc as o
    return o


@pytest.fixture(scope="module")
def unit(amd):
    s = amd.BatchSim(num_envs=1, num_agents=1)
    yield s
    s.close()


# ---------------------------------------------------------------------------- (f)-3 device RNG
def test_noise_stream_is_numpys_bit_for_bit(unit):
    """rng.normal(0., 0.01, 1080) x 10^4 for the seeds of the golden: every double, the ziggurat
    tail samples, and the generator state afterwards (laser_models.py:450-452)"""
    g = gold("noise_stream")
    rows, B, std = int(g["rows"]), int(g["beams"]), float(g["std"])
    for s in [int(v) for v in g["seeds"]]:
        x, st = unit.noise_rows_batch(s, rows, B, std)
        assert np.array_equal(x[:4], g["first_%d" % s]) and np.array_equal(x[-2:], g["last_%d" % s])
        tp = g["tail_pos_%d" % s]
        assert np.array_equal(x[tp[:, 0], tp[:, 1]], g["tail_val_%d" % s])
        assert hashlib.sha256(x.tobytes()).hexdigest() == str(g["sha256_%d" % s])
        assert st == (int(g["state_%d" % s][0]) << 64) | int(g["state_%d" % s][1])
        gen = np.random.Generator(np.random.PCG64(s))
        assert np.array_equal(x, gen.normal(0., std, size=(rows, B)))


@pytest.mark.parametrize("B", [1, 63, 64, 65, 271, 4096])
def test_noise_stream_other_widths_and_seeds(unit, B):
    rows = max(3, 60000 // B)
    for seed in (5 + B, 2 ** 40 + B, 2 ** 64 - 1 - B):
        x, st = unit.noise_rows_batch(seed, rows, B, 0.5)
        gen = np.random.Generator(np.random.PCG64(seed))
        assert np.array_equal(x, gen.normal(0., 0.5, size=(rows, B)))
        assert st == gen.bit_generator.state['state']['state']


def _pair(amd, E, A, beams=1080, **kw):
    img, res, origin = load_map_image("example_map")
    s = amd.BatchSim(num_envs=E, num_agents=A, num_beams=beams, **kw)
    s.set_map_image(img, res, origin)
    return s


def _actions(rng, n):
    return np.stack([rng.uniform(-0.2, 0.2, n), rng.uniform(2.0, 6.0, n)], axis=1)


@pytest.mark.parametrize("cache_rows", [0, 7])
def test_device_rng_step_equals_table_step(amd, cache_rows):
    """the step with the device generator == the step with NumPy's rows uploaded as a table, bit for
    bit, through mask resets (every agent restarts the stream) and — cache_rows=7 — through episodes
    longer than the row cache (k_noise_rows continues from the carried stream position)"""
    E, A, T = 24, 2, 90
    a = _pair(amd, E, A); b = _pair(amd, E, A)
    a.set_noise_rng(12345, 0.01, cache_rows=cache_rows)
    b.set_noise_table(np.random.default_rng(12345).normal(0., 0.01, size=(T + 2, 1080)))
    poses = bench_start_poses(E, A)
    a.reset(poses); b.reset(poses)
    rng = np.random.default_rng(1)
    for t in range(T):
        if t % 20 == 0:
            act = _actions(rng, E * A)
        a.step(act); b.step(act)
        oa = a.get("scans", "state", "collisions", "in_collision", "step_count"); ob = b.get("scans", "state", "collisions", "in_collision", "step_count")
        for k in oa:
            assert np.array_equal(oa[k], ob[k]), (k, t)
        if t % 25 == 24:
            mask = (rng.random(E) < 0.4).astype(np.uint8)
            a.reset(poses, mask); b.reset(poses, mask)
        if t == 60:
            a.reset(poses); b.reset(poses)
    assert oa["step_count"].max() > 7
    a.close(); b.close()


def test_sim_rollout_golden_with_device_rng(amd):
    """the 260-step 2-agent Simulator.step trajectory captured from the Python reference with its
    seed-12345 noise (tests/golden/sim_rollout.npz), noise drawn on the device"""
    g = gold("sim_rollout")
    img, res, origin = load_map_image("example_map")
    T = g["actions"].shape[0]
    s = amd.BatchSim(dict(zip(amd._ffi.PARAM_KEYS, g["params"])), num_envs=1, num_agents=2)
    s.set_map_image(img, res, origin)
    s.set_noise_rng(12345, 0.01, cache_rows=100)   # the trajectory outlives the cache
    s.reset(g["start"])
    full = {int(t): g["scans_t%d" % t] for t in g["full_steps"]}
    for t in range(T):
        s.step(g["actions"][t])
        o = s.get("scans", "state", "collisions", "collision_idx", "in_collision")
        assert np.array_equal(o["collisions"], g["collisions"][t]) and np.array_equal(o["in_collision"], g["in_collision"][t]), t
        assert rel_err(o["state"], g["states"][t]) < 1e-9 and rel_err(o["scans"][:, ::8], g["scans_sub8"][t]) < 1e-9, t
        if t in full:
            assert rel_err(o["scans"], full[t]) < 1e-9
    s.close()


def test_per_agent_noise_streams(amd):
    """extension: a stream per agent.  Single-agent envs are independent, so agent i of the
    per-agent run must equal agent i of a run where everybody uses seed i's stream."""
    E, T = 5, 40
    seeds = [3, 12345, 2 ** 33 + 1, 0, 99]
    poses = bench_start_poses(E, 1)
    rng = np.random.default_rng(2)
    acts = [_actions(rng, E) for _ in range(T)]
    s = _pair(amd, E, 1); s.set_noise_rng(None, 0.01, per_agent_seeds=seeds); s.reset(poses)
    per = []
    for t in range(T):
        s.step(acts[t]); per.append(s.get("scans", "state"))
        if t == 20:
            s.reset(poses, np.array([1, 0, 1, 0, 0], dtype=np.uint8))
    s.close()
    for i, sd in enumerate(seeds):
        r = _pair(amd, E, 1); r.set_noise_rng(sd, 0.01); r.reset(poses)
        for t in range(T):
            r.step(acts[t]); o = r.get("scans", "state")
            assert np.array_equal(o["scans"][i], per[t]["scans"][i]) and np.array_equal(o["state"][i], per[t]["state"][i]), (i, t)
            if t == 20:
                r.reset(poses, np.array([1, 0, 1, 0, 0], dtype=np.uint8))
        r.close()


def test_simulator_default_noise_is_the_device_stream(amd):
    """Simulator(seed) draws on the device by default; noise_mode='table' (NumPy rows uploaded) is the A/B"""
    from f1tenth_gym_amd import Simulator, DEFAULT_PARAMS
    mp = os.path.join(ROOT, "tests", "golden", "maps", "example_map.yaml")
    outs = []
    for mode in ("device", "table"):
        sim = Simulator(DEFAULT_PARAMS, 2, 12345, noise_mode=mode)
        sim.set_map(mp, ".png")
        sim.reset(bench_start_poses(1, 2))
        o = [sim.step(np.array([[0.1, 3.0], [-0.1, 4.0]])) for _ in range(12)][-1]
        outs.append(np.stack(o['scans']))
        sim.batch.close()
    assert np.array_equal(outs[0], outs[1])


# ---------------------------------------------------------------------------- env groups
@pytest.mark.parametrize("A,groups", [(2, 2), (4, 2), (3, 2), (1, 2), (2, 5)])     # (more than two blocks: experimental build)
def test_env_groups_equal_single_block(amd, A, groups):
    """G env blocks on their own streams (pair tests fused into k_integrate for A = 2 / 4, in line
    otherwise) == one block with k_collide on the side stream: every array bit-identical, incl.
    through the fused re-seat, device-side resets and host read-backs in between"""
    E, T = 150, 70
    a = _pair(amd, E, A, step_groups=1); b = _pair(amd, E, A, step_groups=groups)
    poses = bench_start_poses(E, A)
    rng = np.random.default_rng(4)
    for s in (a, b):
        s.set_noise_rng(12345, 0.01)
        s.reset(poses)
    da = [s.device_array((E * A, 2)) for s in (a, b)]
    st = [s.device_array((E * A, 3)) for s in (a, b)]
    for d in st:
        d.upload(poses)
    for t in range(T):
        if t % 10 == 0:
            act = _actions(rng, E * A)
            for d in da:
                d.upload(act)
        if t == 20:
            for s, d in zip((a, b), st):
                s.set_auto_reseat(d, 0, None)
        for s, d in zip((a, b), da):
            s.step_device(d)
        if t % 9 == 8 or t == T - 1:
            oa = a.get("scans", "state", "collisions", "collision_idx", "in_collision", "step_count", "agent_poses")
            ob = b.get("scans", "state", "collisions", "collision_idx", "in_collision", "step_count", "agent_poses")
            for k in oa:
                assert np.array_equal(oa[k], ob[k]), (k, t)
        if t == 40:
            mask = (rng.random(E) < 0.3).astype(np.uint8)
            a.reset(poses, mask); b.reset(poses, mask)
    a.close(); b.close()


def test_env_groups_vs_oracle(amd, orc):
    """grouped stepping (the default for small batches) against the oracle on the bench inputs"""
    E, A, T = 64, 2, 120
    img, res, origin = load_map_image("example_map")
    dt, _, _ = oracle_map_dt("example_map")
    noise = np.random.default_rng(12345).normal(0., 0.01, size=(T + 1, 1080))
    s = _pair(amd, E, A, step_groups=2); s.set_noise_rng(12345, 0.01)
    ref = orc.SimOracle(E, A); ref.set_map_dt(dt, res, origin); ref.set_noise(noise)
    poses = bench_start_poses(E, A)
    s.reset(poses); ref.reset(poses)
    rng = np.random.default_rng(0)
    for t in range(T):
        if t % 20 == 0:
            act = _actions(rng, E * A)
        s.step(act); ref.step(act, 8)
        if t % 10 == 9:
            mask = (ref.collisions.reshape(E, A)[:, 0] != 0).astype(np.uint8)
            s.reset(poses, mask); ref.reset(poses, mask)
        if t % 5 == 0 or t == T - 1:
            o = s.get("scans", "state", "collisions", "in_collision")
            assert np.array_equal(o["collisions"], ref.collisions) and np.array_equal(o["in_collision"], ref.in_collision), t
            assert rel_err(o["state"], ref.state) < NORTH_STAR and rel_err(o["scans"], ref.scans) < NORTH_STAR, t
    s.close()


# ---------------------------------------------------------------------------- bench.py's L-bar
def test_lookup_counter_equals_oracle(amd, orc):
    """f110_scan_lookup_count == the oracle's count of distance-table lookups (laser_models.py
    :129-143) over the same steps — the L-bar of the roofline's algorithmic bytes"""
    for beams, E, A, T in ((1080, 16, 2, 30), (4096, 4, 2, 8)):
        img, res, origin = load_map_image("example_map")
        dt, _, _ = oracle_map_dt("example_map")
        s = _pair(amd, E, A, beams=beams)
        ref = orc.SimOracle(E, A, num_beams=beams); ref.set_map_dt(dt, res, origin)
        poses = bench_start_poses(E, A)
        s.reset(poses); ref.reset(poses)
        rng = np.random.default_rng(0)
        act = _actions(rng, E * A)
        s.scan_lookup_count(enable=True, read=True)
        l0 = ref.lookups
        for t in range(T):
            s.step(act); ref.step(act, 8)
        got = s.scan_lookup_count(enable=False)
        want = ref.lookups - l0
        if beams == 1080:
            assert got == want, (got, want)
        else:
            # more beams than table directions: the kernel marches each DISTINCT direction once
            # (the count bench.py prices config 5 with); the oracle marches every beam
            assert 0 < got < want, (got, want)
        assert s.scan_lookup_count() == 0   # counting is off again
        s.close()


# ---------------------------------------------------------------------------- BASELINE configs[4]
def test_config5_step_default_layout_vs_oracle(amd, orc):
    """65536 x 4096-beam shape at test size: example_map tiled 2x2 (3200 x 3200 cells), 4096 beams,
    DEFAULT (padded) layout, the dedupe pass + k_expand_beams, 8 envs x 2 agents x 40 steps with
    resets: flags exact, scans within 1e-12 (bit-equal outside the opponent windows)"""
    img, res, origin = load_map_image("example_map")
    dt, _, _ = oracle_map_dt("example_map")
    big_img = np.tile(img, (2, 2))
    E, A, T, B = 8, 2, 40, 4096
    s = amd.BatchSim(num_envs=E, num_agents=A, num_beams=B)
    s.set_map_image(big_img, res, origin)
    big_dt = s.get_map_dt()
    assert big_dt.shape == (3200, 3200)
    assert np.array_equal(big_dt, orc.map_dt_from_image(big_img, res))
    noise = np.random.default_rng(12345).normal(0., 0.01, size=(T + 1, B))
    s.set_noise_rng(12345, 0.01)
    ref = orc.SimOracle(E, A, num_beams=B); ref.set_map_dt(big_dt, res, origin); ref.set_noise(noise)
    poses = bench_start_poses(E, A)
    s.reset(poses); ref.reset(poses)
    rng = np.random.default_rng(0)
    for t in range(T):
        if t % 20 == 0:
            act = _actions(rng, E * A)
        s.step(act); ref.step(act, 8)
        o = s.get("scans", "state", "collisions", "in_collision")
        assert np.array_equal(o["collisions"], ref.collisions) and np.array_equal(o["in_collision"], ref.in_collision), t
        assert rel_err(o["state"], ref.state) < NORTH_STAR
        # map ranges + noise are bit-identical on identical poses (uploaded tables only); the beams the
        # opponent ray-cast rewrote went through device sin/cos (an ulp) — hence 1e-12, not array_equal
        assert rel_err(o["scans"], ref.scans) < (1e-12 if np.array_equal(o["state"], ref.state) else NORTH_STAR), t
        assert np.mean(o["scans"] != ref.scans) < 0.05
        if t % 8 == 7:
            mask = (ref.collisions.reshape(E, A)[:, 0] != 0).astype(np.uint8)
            s.reset(poses, mask); ref.reset(poses, mask)
    s.close()


# ---------------------------------------------------------------------------- the big shipped maps
@pytest.mark.parametrize("name", ["vegas", "stata_basement"])
def test_shipped_big_maps_scan_and_step(amd, orc, name):
    """vegas (the reference's default map, 54 MB table) and stata_basement: device EDT == oracle EDT,
    scans + terminating cells bit-exact on random free-space poses, and a short step roll-out"""
    from f1tenth_gym_amd.core import load_map_files
    import f1tenth_gym_amd
    mp = os.path.join(os.path.dirname(f1tenth_gym_amd.__file__), "maps", name + ".yaml")
    img, res, origin = load_map_files(mp, ".png")
    dt = orc.map_dt_from_image(img, res)
    E, A, T = 12, 2, 25
    s = amd.BatchSim(num_envs=E, num_agents=A)
    s.set_map_image(img, res, origin)
    assert np.array_equal(s.get_map_dt(), dt)
    H, W = dt.shape
    rng = np.random.default_rng(9)
    free = np.argwhere(dt > 0.4)
    pick = free[rng.choice(len(free), E, replace=False)]
    c, sn = np.cos(origin[2]), np.sin(origin[2])
    lx, ly = (pick[:, 1] + 0.5) * res, (pick[:, 0] + 0.5) * res
    wx, wy = origin[0] + c * lx - sn * ly, origin[1] + sn * lx + c * ly
    poses = np.stack([wx, wy, rng.uniform(0, 6.28, E)], axis=1)
    so = orc.ScanOracle(1080, 4.7); so.set_map_dt(dt, res, origin)
    ranges, hits = s.scan_batch(poses, want_hits=True)
    for i in range(E):
        r, hcell = so.scan(poses[i], want_hits=True)
        assert np.array_equal(ranges[i], r) and np.array_equal(hits[i], hcell), i
    both = np.repeat(poses, A, axis=0)
    both[1::2, 0] += 0.9 * np.cos(both[1::2, 2] + 2.5); both[1::2, 1] += 0.9 * np.sin(both[1::2, 2] + 2.5)
    ref = orc.SimOracle(E, A); ref.set_map_dt(dt, res, origin)
    s.reset(both); ref.reset(both)
    for t in range(T):
        act = np.stack([rng.uniform(-0.3, 0.3, E * A), rng.uniform(1.0, 6.0, E * A)], axis=1)
        s.step(act); ref.step(act, 8)
        o = s.get("scans", "state", "collisions", "in_collision")
        assert np.array_equal(o["collisions"], ref.collisions) and np.array_equal(o["in_collision"], ref.in_collision), t
        assert rel_err(o["state"], ref.state) < NORTH_STAR and rel_err(o["scans"], ref.scans) < NORTH_STAR, t
    s.close()


# ---------------------------------------------------------------------------- RCCL gather, world size 1
def test_comm_all_gather_world_size_one(amd):
    """f110_comm_* (the optional observation gather of BASELINE configs[3]) on one rank: RCCL is
    resolved and initialised, the gathered block equals the rank's own scans"""
    E, A = 16, 2
    s = _pair(amd, E, A)
    poses = bench_start_poses(E, A)
    s.reset(poses)
    s.comm_init(1, 0, amd.BatchSim.comm_unique_id())
    d_all = s.device_array((1, E * A, 1080))
    rng = np.random.default_rng(0)
    for t in range(3):
        s.step(_actions(rng, E * A))
        s.comm_all_gather_scans(d_all)
        assert np.array_equal(d_all.download()[0], s.get("scans")["scans"])
    d_all.free()
    s.close()


def test_comm_overlapped_gather_double_buffered(amd):
    """f110_comm_set_overlap: the gather of step t runs on its own stream beside step t+1, which
    fills the second scans buffer.  World size 1: every gathered block must equal the scans of the
    step it was issued after, also when the next step has already overwritten... the other buffer."""
    E, A, T = 64, 2, 9
    s = _pair(amd, E, A); ref = _pair(amd, E, A)
    poses = bench_start_poses(E, A)
    for x in (s, ref):
        x.set_noise_rng(12345, 0.01); x.reset(poses)
    s.comm_init(1, 0, amd.BatchSim.comm_unique_id())
    s.comm_set_overlap(True)
    recv = [s.device_array((1, E * A, 1080)) for _ in range(2)]
    rng = np.random.default_rng(0)
    want = []
    for t in range(T):
        act = _actions(rng, E * A)
        s.step(act); ref.step(act)
        s.comm_all_gather_scans(recv[t % 2])
        want.append(ref.get("scans")["scans"])
        if t >= 1:   # consume the PREVIOUS gather while this step's gather is in flight
            assert np.array_equal(recv[(t - 1) % 2].download()[0], want[t - 1]), t
        assert np.array_equal(s.get("scans")["scans"], want[t]), t   # observations unaffected by the buffering
    assert np.array_equal(recv[(T - 1) % 2].download()[0], want[T - 1])
    s.comm_set_overlap(False)
    s.step(act); ref.step(act)
    assert np.array_equal(s.get("scans")["scans"], ref.get("scans")["scans"])
    for d in recv:
        d.free()
    s.close(); ref.close()


# ---------------------------------------------------------------------------- fuzzers, bounded seeds
def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


NESTED = bool(os.environ.get("F110_NESTED_SUITE"))   # the lab build's re-run of the suite runs a few seeds of every chunk


@pytest.mark.parametrize("first", range(0, 300, 50))
def test_fuzz_parity_bounded_seeds(amd, first):
    """tools/debug/fuzz_parity.py, seeds 0 .. 299 in the driver-run suite (round 6; 12 before, 1 500 by hand in round 5): the HIP step
    against the CPU oracle over random maps / agents / beams / fov / integrator / lidar offset / layout / noise / launch geometry /
    constructor arguments / origin yaw / time step; ~0.11 s per seed on the box"""
    fz = _load(os.path.join(ROOT, "tools", "debug", "fuzz_parity.py"), "fuzz_parity")
    # flags and step counters exact, floats at 1e-9; a seed whose random actions blow a car's state
    # past 1e6 is compared up to that step (beyond it ulp differences amplify without bound)
    seeds = range(first, first + (2 if NESTED else 50))
    bad = [sd for sd in seeds if not fz.run(sd, verbose=False, stop_when_diverged=True)]
    assert not bad, bad


@pytest.mark.parametrize("first", range(0, 12, 4))
def test_fuzz_units_seeds(amd, first):
    """tools/debug/fuzz_units.py, seeds 0 .. 11 (round 6; seed 0 before): every unit entry point of the C ABI against the oracle"""
    import re
    for seed in range(first, first + (1 if NESTED else 4)):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug", "fuzz_units.py"), str(seed)], stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, text=True, timeout=900, cwd=ROOT)
        assert out.returncode == 0, out.stdout[-2000:]
        for line in out.stdout.splitlines():
            m = re.search(r"(mismatches|bad cases|bad poses) (\d+)", line)
            if m:
                assert int(m.group(2)) == 0, (seed, line)
            if "exact" in line and ("True" in line or "False" in line):
                assert "False" not in line, (seed, line)
            for r in re.findall(r"rel ([0-9.e+-]+|inf|nan)", line):
                assert float(r) < 1e-9, (seed, line)


# ---------------------------------------------------------------------------- import-level drop-in
def test_f110_gym_alias_make_reset_step(amd):
    """the reference's usage (examples/waypoint_follow.py:272-285) with its own package name"""
    import f110_gym
    from f110_gym.envs.base_classes import Integrator
    mp = os.path.join(ROOT, "tests", "golden", "maps", "example_map")
    env = f110_gym.make('f110_gym:f110-v0', map=mp, map_ext='.png', num_agents=1, timestep=0.01, integrator=Integrator.RK4)
    obs, step_reward, done, info = env.reset(np.array([[0.7, 0.0, 1.37079632679]]))
    assert not done and obs['scans'][0].shape == (1080,) and step_reward == 0.01
    for _ in range(5):
        obs, step_reward, done, info = env.step(np.array([[0.0, 2.0]]))
    assert obs['linear_vels_x'][0] > 0.1 and 'checkpoint_done' in info
    env.sim.batch.close()


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` must not run a smaller job under the N label"""
    from f1tenth_gym_amd import _ffi
    n = _ffi.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode != 0 and "refusing" in out.stderr and not out.stdout.strip()








def test_memory_flat_in_long_auto_reset_loop(amd):
    """(f)-3's reason to exist: a training-style loop must not grow memory with the step count (the
    round-1 host noise table grew 8.6 KB per step of the longest episode).  4000 steps, row cache of 32
    rows so that live episodes run far past it; device free memory identical at 1000 and 4000 steps."""
    from _util import MAPS
    E = 8
    env = amd.F110VecEnv(E, auto_reset=True, device_logic=True, obs_fields=("collisions",), map=os.path.join(MAPS, "example_map"), map_ext=".png", num_agents=2)
    b = env.sim.batch
    b.set_noise_rng(12345, 0.01, cache_rows=32)
    env.reset(bench_start_poses(E, 2).reshape(E, 2, 3))
    act = np.tile(np.array([[0.0, 1.0]]), (E, 2, 1))
    free_at = {}
    longest = 0
    for t in range(4000):
        env.step(act)
        if t in (999, 3999):
            free_at[t] = b.device_mem_info()[0]
            longest = max(longest, int(b.get("step_count")["step_count"].max()))
    assert longest > 32, "no episode outlived the row cache"
    assert free_at[999] == free_at[3999], free_at
    b.close()


def test_vec_env_single_env_host_path(amd):
    """F110VecEnv(num_envs=1, num_agents=2) on the HOST logic path keeps the batched layout (leading env
    axis) — round 1 inferred the layout from num_envs == 1 and raised on reset — and equals env 0 of a
    2-env run; a partial reset re-seats without stepping anybody"""
    from _util import MAPS
    kw = dict(map=os.path.join(MAPS, "example_map"), map_ext='.png', num_agents=2, seed=12345)
    one = amd.F110VecEnv(1, **kw)
    two = amd.F110VecEnv(2, **kw)
    p = bench_start_poses(2, 2).reshape(2, 2, 3)
    o1, _, d1, _ = one.reset(p[:1]); o2, _, d2, _ = two.reset(p)
    assert o1['scans'].shape == (1, 2, 1080) and d1.shape == (1,)
    rng = np.random.default_rng(1)
    for t in range(15):
        act = np.stack([rng.uniform(-0.2, 0.2, (2, 2)), rng.uniform(1, 5, (2, 2))], axis=2)
        o1, _, d1, i1 = one.step(act[:1]); o2, _, d2, i2 = two.step(act)
        for k in ("scans", "poses_x", "poses_y", "poses_theta", "linear_vels_x", "collisions", "lap_times"):
            assert np.array_equal(o1[k][0], o2[k][0]), (k, t)
    # partial reset: env 1 re-seated, env 0 untouched, nobody stepped
    before = two.sim.batch.get("state", "step_count")
    o, _, d, _ = two.reset(p, env_mask=np.array([0, 1], dtype=np.uint8))
    after = two.sim.batch.get("state", "step_count")
    assert np.array_equal(before["state"][:2], after["state"][:2]) and np.array_equal(before["step_count"][:2], after["step_count"][:2])
    assert np.array_equal(after["step_count"][2:], [0, 0]) and np.allclose(after["state"][2:, 0], p[1, :, 0])
    assert not d[1]
    one.sim.batch.close(); two.sim.batch.close()


def test_bench_two_launched_ranks_on_one_gpu():
    """the driver's N > 1 form — N processes with RANK / LOCAL_RANK / WORLD_SIZE set — with real GPU steps:
    two ranks share device 0 (F110_BENCH_DEVICE, a testing aid), each steps its own env shard, rank 0
    prints one line for the whole job"""
    import json
    import socket
    import random
    port = None
    for _ in range(200):      # below the kernel's ephemeral range: a port from bind(0) can go to somebody's outgoing connection meanwhile
        cand = random.randint(20000, 29999)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", cand))
            port = cand
            break
        except OSError:
            pass
        finally:
            s.close()
    assert port is not None
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   F110_BENCH_DEVICE="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5",
                                       "--agents", "4096", "--no-cpu-baseline", "--no-gather-legs"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not outs[1][0].strip()
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["agents_total"] == 8192 and d["scaling"] == "weak"
    assert abs(d["value"] - 8192 * 30 / (d["ms_per_step"] * 30e-3)) < 1e-6 * d["value"]
    assert "roofline" in d and 0.0 < d["roofline"]["frac"] < 1.0 and d["roofline"]["lookups_per_ray"] > 5.0




# ---------------------------------------------------------------------------- longest-first task order
@pytest.mark.parametrize("thr", ["4", "96"])
def test_longest_first_task_order_is_invisible(amd, monkeypatch, thr):
    """small batches: tasks whose longest ray was long in the previous step are served by the first blocks
    of the next scan launch and skipped by the normal blocks (TaskSched).  Forced on — with a low threshold,
    so that most tasks go through the list, and with the default one — against forced off: not a bit may
    change, through resets, re-seat arming, lookup counting (which suspends the ordering) and 70 steps"""
    E, A, T = 300, 2, 70
    monkeypatch.setenv("F110_EXP", "task_order=0")   # switches of the experimental build
    a = _pair(amd, E, A)
    monkeypatch.setenv("F110_EXP", "task_order=1,task_thr=%s" % thr)
    b = _pair(amd, E, A)
    poses = bench_start_poses(E, A)
    rng = np.random.default_rng(8)
    for s in (a, b):
        s.set_noise_rng(12345, 0.01)
        s.reset(poses)
    st = [s.device_array((E * A, 3)) for s in (a, b)]
    for d in st:
        d.upload(poses)
    for t in range(T):
        if t % 10 == 0:
            act = _actions(rng, E * A)
        if t == 30:
            for s, d in zip((a, b), st):
                s.set_auto_reseat(d, 0, None)
        if t == 50:
            for s in (a, b):
                s.scan_lookup_count(enable=True, read=True)
        if t == 55:
            assert a.scan_lookup_count(enable=False) == b.scan_lookup_count(enable=False)
        a.step(act); b.step(act)
        oa = a.get("scans", "state", "collisions", "in_collision", "step_count"); ob = b.get("scans", "state", "collisions", "in_collision", "step_count")
        for kk in oa:
            assert np.array_equal(oa[kk], ob[kk]), (kk, t)
        if t == 45:
            mask = (rng.random(E) < 0.3).astype(np.uint8)
            a.reset(poses, mask); b.reset(poses, mask)
    a.close(); b.close()
