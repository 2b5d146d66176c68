"""GPU tests added in round 3 (-m gpu): the workgroup-flattened finalize, per-env tracks with more beams
than table directions and with 8 slots, the whole-observation RCCL gather at world size 1, BASELINE
configs[4] at FULL size through size-independent properties, the full reference observation key set of
F110VecEnv, object lifetimes (DeviceArray, pinned blocks), seed=None, and — once, from the default run —
the whole GPU suite again on the experimental build of the library."""
import gc
import os
import subprocess
import sys

import numpy as np
import pytest

from _util import load_map_image, oracle_map_dt, bench_start_poses, raceline, rel_err

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


def _actions(rng, n):
    return np.stack([rng.uniform(-0.3, 0.3, n), rng.uniform(0.5, 7.0, n)], axis=1)


def _sim(amd, E, A, **kw):
    s = amd.BatchSim(num_envs=E, num_agents=A, **kw)
    s.set_map_image(*load_map_image("example_map"))
    return s


ALL = ("scans", "state", "collisions", "collision_idx", "in_collision", "step_count")


# ---------------------------------------------------------------------------- both builds
def test_gpu_suite_on_the_experimental_build():
    """the product library has one step dispatch per case and no switches; everything that
    was measured and not adopted lives in libf110_hip_exp.so.  The default run (product library) launches
    the whole -m gpu suite once more against that build, where the A/B tests that skip here run for real."""
    from f1tenth_gym_amd import _ffi
    if _ffi.VARIANT == "experimental" or os.environ.get("F110_NESTED_SUITE"):
        pytest.skip("already inside the experimental-build run")
    env = dict(os.environ, F110_LIB_VARIANT="experimental", F110_NESTED_SUITE="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=3000)
    tail = out.stdout[-3000:]
    assert out.returncode == 0, tail
    assert " passed" in tail and " failed" not in tail, tail
    # the count is part of the assertion: a nested collection problem that still exits 0 with "1 passed" would
    # otherwise be invisible.  In the lab build nothing of the suite may skip except this test and the two tests that
    # need several devices; everything the product run skipped as LAB_ONLY must have run here.
    import re
    m = re.search(r"(\d+) passed(?:, (\d+) skipped)?", tail)
    assert m, tail
    passed, skipped = int(m.group(1)), int(m.group(2) or 0)
    collected = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-m", "gpu", "--collect-only", "-q", "-p", "no:cacheprovider"],
                               env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600).stdout
    n_tests = int(re.search(r"(\d+)(?:/\d+)? tests collected", collected).group(1))
    assert passed + skipped == n_tests, (passed, skipped, n_tests, tail)
    assert passed >= 150 and skipped <= 10, (passed, skipped, tail)


def test_product_library_refuses_the_lab(amd):
    from f1tenth_gym_amd import _ffi
    if _ffi.VARIANT == "experimental":
        assert _ffi.lib().f110_is_experimental() == 1
        s = amd.BatchSim(num_envs=2, num_agents=2, exp={"integrate_fan": 1})
        with pytest.raises(ValueError):
            s.exp_set("no_such_switch", 1)
        with pytest.raises(ValueError):
            s.exp_set("finalize_flat", 1)       # a switch retired in round 5
        for key in ("scan_stream", "spec_from", "finalize_wave", "pad_tiled", "scan_nt"):     # ... and round 6's retirements (the stop rule)
            with pytest.raises(ValueError, match="retired in round 6"):
                s.exp_set(key, 1)
        s.close()
    else:
        assert _ffi.lib().f110_is_experimental() == 0
        for kw in ({"step_groups": 3}, {"exp": {"collide_mode": 0}}, {"exp": {"integrate_fan": 1}}):
            with pytest.raises(_ffi.ExperimentalOnly):
                amd.BatchSim(num_envs=2, num_agents=2, **kw)
    # retired in round 5 (measured slower in rounds 1-4, numbers in DESIGN_HISTORY.md): refused by BOTH builds
    for kw in ({"map_layout": 1}, {"map_layout": 2}, {"map_layout": 4}, {"step_graph": 1}):
        with pytest.raises(ValueError) as ei:
            amd.BatchSim(num_envs=2, num_agents=2, **kw)
        assert "retired in round 5" in str(ei.value)




@pytest.mark.parametrize("E,A,integrator,lidar_dist", [(300, 2, 1, 0.0), (37, 3, 1, 0.275), (65, 1, 2, 0.0)])
def test_integrate_in_two_waves_is_invisible(amd, E, A, integrator, lidar_dist):
    """k_integrate_duo (a second wave walks steer / velocity one RK4 stage ahead and leaves the low-speed branch's
    tan / cos in LDS) against the one-wave kernel: the same operations on the same values, so not a bit may differ —
    from standstill (every lane in the low-speed branch) through the |v| = 0.5 crossing, with resets, agent counts
    that do not fill the last workgroup, Euler, an offset lidar"""
    T = 60
    kw = dict(integrator=integrator, lidar_dist=lidar_dist)   # (1 = RK4, 2 = Euler: f110.h)
    a = _sim(amd, E, A, exp={"integrate_duo": 0}, **kw); b = _sim(amd, E, A, exp={"integrate_duo": 1}, **kw)
    poses = bench_start_poses(E, A)
    rng = np.random.default_rng(33)
    for s in (a, b):
        s.set_noise_rng(12345, 0.01); s.reset(poses)
    for t in range(T):
        if t % 6 == 0:
            act = np.stack([rng.uniform(-0.4, 0.4, E * A), rng.uniform(-1.0, 4.0, E * A)], axis=1)   # slow: many lanes stay near |v| = 0.5
        a.step(act); b.step(act)
        oa, ob = a.get(*ALL), b.get(*ALL)
        for kk in oa:
            assert np.array_equal(oa[kk], ob[kk]), (kk, t)
        if t == 30:
            mask = (rng.random(E) < 0.4).astype(np.uint8)
            a.reset(poses, mask); b.reset(poses, mask)
    v = a.get("state")["state"][:, 3]
    assert (np.abs(v) < 0.5).any() and (np.abs(v) >= 0.5).any()      # both branches were live at the end
    a.close(); b.close()


@pytest.mark.parametrize("E,A", [(130, 3), (101, 4), (37, 5), (9, 8), (1, 3), (7, 9), (5, 12), (3, 16)])
def test_finalize_multi_is_invisible(amd, E, A):
    """envs of 3..16 agents: pair tests, opponent windows and the ray-cast inside ONE finalize kernel (k_finalize_multi:
    every ordered pair of an env a record, windows flattened, overlapping opponents settled by an integer atomicMin on
    the range's bit pattern) against round 1's form (k_collide on the side stream + k_finalize): not a bit may
    differ — cars that start nose to tail, crash into each other and into walls, re-seat, partial last workgroups"""
    T = 80
    a = _sim(amd, E, A, exp={"collide_mode": 0}); b = _sim(amd, E, A)
    poses = bench_start_poses(E, A, gap_wp=4)      # close: opponents fill each other's windows, pairs collide early
    rng = np.random.default_rng(40 + A)
    for s in (a, b):
        s.set_noise_rng(12345, 0.01); s.reset(poses)
    st = [s.device_array((E * A, 3)) for s in (a, b)]
    for d in st:
        d.upload(poses)
    n_pair = n_wall = n_multi = 0
    for t in range(T):
        if t % 8 == 0:
            act = _actions(rng, E * A)
        if t == 40:
            for s, d in zip((a, b), st):
                s.set_auto_reseat(d, 0, None)
        a.step(act); b.step(act)
        oa, ob = a.get(*ALL), b.get(*ALL)
        for kk in oa:
            assert np.array_equal(oa[kk], ob[kk]), (kk, t, A)
        n_pair += int((oa["collision_idx"] >= 0).sum()); n_wall += int(oa["in_collision"].sum())
        n_multi += int(((oa["collision_idx"] >= 0).reshape(E, A).sum(axis=1) > 2).sum())
        if t == 25:
            mask = (rng.random(E) < 0.5).astype(np.uint8)
            a.reset(poses, mask); b.reset(poses, mask)
    assert n_pair > 0 and (n_wall > 0 or E < 5), (n_pair, n_wall)
    a.close(); b.close()


def _set_trace(s, ptr):
    s.exp_set("scan_trace_hi", int(np.array(ptr >> 32, dtype=np.uint32).view(np.int32)))
    s.exp_set("scan_trace_lo", int(np.array(ptr & 0xffffffff, dtype=np.uint32).view(np.int32)))


def test_scan_timeline_probe_and_list_switches_are_invisible(amd):
    """the wave-by-wave timeline of the scan launch (tools/debug/scan_timeline.py: every wave stamps begin / end /
    phase clocks, its CU and its lock-step samples into a caller-owned buffer) and the longest-first list's capacity
    and walking order change no result; the records it leaves are coherent"""
    E, A, T = 400, 2, 24
    B, tpa = 1080, 17
    a = _sim(amd, E, A, exp={"task_order": 1})
    b = _sim(amd, E, A, exp={"task_order": 1, "task_thr": 8, "task_cap_div": 2, "task_rev": 1})
    poses = bench_start_poses(E, A)
    rng = np.random.default_rng(21)
    for s in (a, b):
        s.set_noise_rng(12345, 0.01); s.reset(poses)
    n_waves = E * A * tpa + E * A * tpa // 2 + 4096
    tr = b.device_array((n_waves, 8), dtype=np.uint64)
    for t in range(T):
        if t % 8 == 0:
            act = _actions(rng, E * A)
        if t == 10:
            tr.upload(np.zeros((n_waves, 8), dtype=np.uint64)); _set_trace(b, tr.ptr)
        if t == 12:
            _set_trace(b, 0)
        a.step(act); b.step(act)
        oa, ob = a.get(*ALL), b.get(*ALL)
        for kk in oa:
            assert np.array_equal(oa[kk], ob[kk]), (kk, t)
    r = tr.download()
    live = r[r[:, 1] != 0]
    assert E * A * tpa * 0.5 < len(live) <= n_waves            # (the last traced launch; listed tasks' normal waves end too)
    begin, end, loop, hdr, ops, marched = (live[:, c].astype(np.int64) for c in (0, 1, 4, 5, 6, 7))
    assert (begin <= loop).all() and (loop <= end).all()
    worked = ops != 0
    assert worked.sum() > E * A * tpa * 0.5
    assert (loop[worked] <= hdr[worked]).all() and (hdr[worked] <= ops[worked]).all() and (ops[worked] <= marched[worked]).all() and (marched[worked] <= end[worked]).all()
    samples = (live[:, 3] & 0xffffffff).astype(np.int64)
    assert 0 < samples[worked].max() < 2000 and samples[worked].mean() > 2
    assert ((live[:, 3] >> 32) != 0).sum() > 0                   # some waves served the list
    assert len(np.unique(live[:, 2])) > 64                       # many distinct (CU, SIMD, slot) ids
    a.close(); b.close()




@pytest.mark.parametrize("probe", [{"scan_occupancy": 4}, {"scan_env_counter": 1}])
def test_fusion_probes_do_not_change_results(amd, probe):
    """the two probes behind DESIGN 4.4's fusion-feasibility numbers (scan kernel at 4 waves/SIMD; per-env
    completion counter) only cost time"""
    E, A, T = 64, 2, 12
    a = _sim(amd, E, A, exp={"task_order": 0}); b = _sim(amd, E, A, exp=dict(probe, task_order=0))
    poses = bench_start_poses(E, A)
    rng = np.random.default_rng(3)
    for s in (a, b):
        s.set_noise_rng(12345, 0.01); s.reset(poses)
    for t in range(T):
        act = _actions(rng, E * A)
        a.step(act); b.step(act)
    oa, ob = a.get(*ALL), b.get(*ALL)
    for kk in oa:
        assert np.array_equal(oa[kk], ob[kk]), kk
    a.close(); b.close()




def test_step_scan_is_the_unit_scan_hit_cells_included(amd, orc):
    """hit-cell exactness was asserted on the unit entry point only (the step kernels compile the cell
    bookkeeping out).  Tie the two: over a roll-out (single-agent envs, no noise, so nothing but the march writes
    the scans) the step's ranges are bit for bit the unit scan's at the poses the step scanned from, and the unit
    scan's ranges and terminating cells at those poses are the oracle's"""
    E, A, T = 96, 1, 40
    img, res, origin = load_map_image("example_map")
    dt, _, _ = oracle_map_dt("example_map")
    s = amd.BatchSim(num_envs=E, num_agents=A)
    s.set_map_image(img, res, origin)
    unit = amd.BatchSim(num_envs=1, num_agents=1)
    unit.set_map_image(img, res, origin)
    so = orc.ScanOracle(1080, 4.7); so.set_map_dt(dt, res, origin)
    poses = bench_start_poses(E, A)
    s.reset(poses)
    rng = np.random.default_rng(4)
    checked = 0
    for t in range(T):
        if t % 10 == 0:
            act = _actions(rng, E * A)
        s.step(act)
        if t % 5 == 4:
            o = s.get("scans", "agent_poses")
            ranges, hits = unit.scan_batch(o["agent_poses"], want_hits=True)
            assert np.array_equal(o["scans"], ranges), t                     # step kernel == unit kernel, every beam
            for i in range(0, E, 4):                                         # the oracle at the very poses the step scanned from
                r_ref, h_ref = so.scan(o["agent_poses"][i], want_hits=True)
                assert np.array_equal(ranges[i], r_ref) and np.array_equal(hits[i], h_ref), (t, i)
                checked += 1080
    assert checked >= 8 * (E // 4) * 1080
    s.close(); unit.close()


# ---------------------------------------------------------------------------- (f)-2: many tracks, any beam count
def _track_poses(rng, env_map, A):
    w = raceline()
    E = len(env_map)
    poses = np.zeros((E, A, 3))
    for e in range(E):
        if env_map[e] % 3 == 0:      # example_map (or a copy of it)
            k = rng.integers(0, w.shape[0]); base = np.array([w[k, 1], w[k, 2], w[k, 3] + np.pi / 2])
        else:
            base = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(0, 6.28)])
        for a in range(A):
            poses[e, a] = base + np.array([rng.uniform(-0.7, 0.7), rng.uniform(-0.7, 0.7), rng.uniform(-0.4, 0.4)])
    return poses.reshape(E * A, 3)


@pytest.mark.parametrize("beams,slots", [(1080, 8), (4096, 3)])
def test_per_env_maps_many_slots_and_beams(amd, beams, slots):
    """f110_set_env_maps with 8 registered tracks (grouped AND interleaved assignment), and with more beams
    than table directions (the refusal of round 2 is gone: k_scan_dirs_agent takes the per-env map record):
    every env evolves bit for bit like a single-map simulator of its track"""
    names = ["example_map", "berlin", "skirk"]
    maps = [load_map_image(names[m % 3]) for m in range(slots)]
    E, A, T = 4 * slots, 2, 25
    rng = np.random.default_rng(8)
    for assign in ("interleaved", "grouped"):
        env_map = (np.arange(E) % slots) if assign == "interleaved" else (np.arange(E) // (E // slots))
        poses = _track_poses(rng, env_map, A)
        multi = amd.BatchSim(num_envs=E, num_agents=A, num_beams=beams)
        multi.set_map_image(*maps[0])
        for m in range(1, slots):
            assert multi.add_map_image(*maps[m]) == m
        multi.set_env_maps(env_map)
        multi.set_noise_rng(12345, 0.01)
        multi.reset(poses)
        singles, sel = [], []
        for m in range(slots):
            idx = np.where(env_map == m)[0]
            ag = (idx[:, None] * A + np.arange(A)[None, :]).reshape(-1)
            s = amd.BatchSim(num_envs=len(idx), num_agents=A, num_beams=beams)
            s.set_map_image(*maps[m]); s.set_noise_rng(12345, 0.01); s.reset(poses[ag])
            singles.append(s); sel.append(ag)
        for t in range(T):
            act = np.stack([rng.uniform(-0.4, 0.4, E * A), rng.uniform(0.5, 8.0, E * A)], axis=1)
            multi.step(act)
            for s, ag in zip(singles, sel):
                s.step(act[ag])
            if t % 6 == 5 or t == T - 1:
                o = multi.get(*ALL)
                for s, ag in zip(singles, sel):
                    q = s.get(*ALL)
                    for key in q:
                        assert np.array_equal(o[key][ag], q[key]), (assign, t, key)
        multi.close()
        for s in singles:
            s.close()


# ---------------------------------------------------------------------------- whole-observation gather
def test_comm_all_gather_obs_world_size_one(amd):
    """f110_comm_all_gather_obs on one rank: the scans and the [7][N] scalar block (poses_x, poses_y,
    poses_theta, linear_vels_x, linear_vels_y = 0, ang_vels_z, collisions) of the step just taken, in the
    step's stream and overlapped with the next step (double-buffered)"""
    E, A, T = 48, 2, 10
    s = _sim(amd, E, A); ref = _sim(amd, E, A)
    poses = bench_start_poses(E, A, gap_wp=3)
    for x in (s, ref):
        x.set_noise_rng(12345, 0.01); x.reset(poses)
    s.comm_init(1, 0, amd.BatchSim.comm_unique_id())
    assert s.comm_info() == (1, 0)
    recv = [(s.device_array((1, E * A, 1080)), s.device_array((1, 7, E * A))) for _ in range(2)]
    rng = np.random.default_rng(0)

    def want_of(x):
        o = x.get("scans", "poses_x", "poses_y", "poses_theta", "linear_vels_x", "ang_vels_z", "collisions")
        return o["scans"], np.stack([o[k] if k != "linear_vels_y" else np.zeros(E * A) for k in amd.BatchSim.OBS_SCALARS])

    for t in range(T):
        act = _actions(rng, E * A)
        s.step(act); ref.step(act)
        s.comm_all_gather_obs(*recv[0])
        ws, wc = want_of(ref)
        assert np.array_equal(recv[0][0].download()[0], ws) and np.array_equal(recv[0][1].download()[0], wc), t
    s.comm_set_overlap(True)
    want = []
    for t in range(T):
        act = _actions(rng, E * A)
        s.step(act); ref.step(act)
        s.comm_all_gather_obs(*recv[t % 2])
        want.append(want_of(ref))
        if t >= 1:
            assert np.array_equal(recv[(t - 1) % 2][0].download()[0], want[t - 1][0]) and np.array_equal(recv[(t - 1) % 2][1].download()[0], want[t - 1][1]), t
    assert np.array_equal(recv[(T - 1) % 2][1].download()[0], want[T - 1][1])
    s.comm_set_overlap(False)
    s.close(); ref.close()


# ---------------------------------------------------------------------------- BASELINE configs[4] at full size
def test_config5_full_size_properties(amd, orc):
    """65 536 agents x 4096 beams on the 3200 x 3200 table (BASELINE configs[4]) at FULL size, through
    size-independent properties: twin envs (same start pose and actions) produce identical rows, a slice of
    32 envs matches the oracle, ranges stay in range — every single-GPU config now has a full-size witness"""
    img, res, origin = load_map_image("example_map")
    big_img = np.tile(img, (2, 2))
    E, A, T, B = 32768, 2, 4, 4096
    s = amd.BatchSim(num_envs=E, num_agents=A, num_beams=B)
    s.set_map_image(big_img, res, origin)
    big_dt = s.get_map_dt()
    s.set_noise_rng(12345, 0.01)
    base = bench_start_poses(783, A).reshape(783, A, 3)
    poses = base[np.arange(E) % 783].reshape(E * A, 3)       # env e and e + 783 are twins
    s.reset(poses)
    rng = np.random.default_rng(1)
    a783 = np.stack([rng.uniform(-0.2, 0.2, (783, A)), rng.uniform(2, 6, (783, A))], axis=2)
    act = a783[np.arange(E) % 783].reshape(E * A, 2)
    noise = np.random.default_rng(12345).normal(0., 0.01, size=(T + 1, B))
    ref = orc.SimOracle(32, A, num_beams=B); ref.set_map_dt(big_dt, res, origin); ref.set_noise(noise); ref.reset(poses[:64])
    for t in range(T):
        s.step(act); ref.step(act[:64], 8)
    o = s.get("state", "collisions", "in_collision")
    stt = o["state"].reshape(E, A, 7)
    twins = np.arange(783, 783 * 3)
    assert np.array_equal(stt[twins], stt[twins % 783])
    assert np.array_equal(o["collisions"][:64], ref.collisions) and np.array_equal(o["in_collision"][:64], ref.in_collision)
    assert rel_err(o["state"][:64], ref.state) < NORTH_STAR
    views = s.device_views()
    rows = views["scans"]
    # scans: 2.1 GB on the device — compare slices (twins, the oracle's 64 agents, range bounds on 4096 agents)
    head = np.empty((783 * 3 * A, B))
    from f1tenth_gym_amd import _ffi
    _ffi.check(_ffi.lib().f110_memcpy_d2h(s._h, head.ctypes.data, rows.ptr, head.nbytes), s._h)
    sc = head.reshape(783 * 3, A, B)
    assert np.array_equal(sc[twins], sc[twins % 783])
    assert rel_err(head[:64], ref.scans) < NORTH_STAR
    assert head.min() > -0.06 and head.max() < 30.06
    s.close()


# ---------------------------------------------------------------------------- drop-in details
def test_vec_env_emits_every_reference_observation_key(amd):
    """docs/api/obv.rst:6-14 / base_classes.py:594-610: ego_idx, scans, poses_x, poses_y, poses_theta,
    linear_vels_x, linear_vels_y, ang_vels_z, collisions (+ lap_times, lap_counts from F110Env.step) — from
    the host path and from the device-logic path alike"""
    want = {"ego_idx", "scans", "poses_x", "poses_y", "poses_theta", "linear_vels_x", "linear_vels_y", "ang_vels_z", "collisions",
            "lap_times", "lap_counts"}
    E, A = 6, 2
    poses = bench_start_poses(E, A).reshape(E, A, 3)
    for device_logic in (False, True):
        env = amd.F110VecEnv(E, map=os.path.join(ROOT, "tests", "golden", "maps", "example_map"), map_ext=".png", num_agents=A,
                             device_logic=device_logic)
        obs, _, _, _ = env.reset(poses)
        assert set(obs) == want, (device_logic, sorted(obs))
        assert obs["linear_vels_y"].shape == (E, A) and not obs["linear_vels_y"].any()
        obs, _, _, _ = env.step(np.tile([0.1, 3.0], (E, A, 1)))
        assert set(obs) == want and obs["scans"].shape == (E, A, 1080)
        env.sim.batch.close()


def test_seed_none_draws_fresh_noise_as_the_reference_does(amd):
    """Simulator(seed=None) must behave like default_rng(None) (base_classes.py:204): noise ON,
    different from run to run — not silently off"""
    E, A = 1, 2
    poses = bench_start_poses(E, A)
    outs = []
    for _ in range(2):
        sim = amd.Simulator(amd.DEFAULT_PARAMS, A, None)
        sim.set_map(os.path.join(ROOT, "tests", "golden", "maps", "example_map.yaml"), ".png")
        sim.reset(poses)
        outs.append(np.array(sim.step(np.zeros((A, 2)))["scans"]))
        sim.batch.close()
    quiet = amd.Simulator(amd.DEFAULT_PARAMS, A, 12345, scan_noise_std=0.0)
    quiet.set_map(os.path.join(ROOT, "tests", "golden", "maps", "example_map.yaml"), ".png")
    quiet.reset(poses)
    clean = np.array(quiet.step(np.zeros((A, 2)))["scans"])
    quiet.batch.close()
    assert not np.array_equal(outs[0], outs[1])
    for o in outs:
        d = o - clean
        assert 0.005 < d.std() < 0.02 and np.abs(d).max() < 0.08


def test_device_arrays_and_pinned_blocks_have_owners(amd):
    """DeviceArray: context manager, free() twice, collected without free(), alive at close(); a pinned block
    stays valid for the arrays that view it after the handle is closed (use-after-free)"""
    s = _sim(amd, 4, 2)
    gc.collect()                     # earlier tests' garbage first, so that the numbers below are this test's
    free0 = s.device_mem_info()[0]
    with s.device_array((1 << 20,)) as d:
        d.upload(np.arange(1 << 20, dtype=np.float64))
        assert d.download()[12345] == 12345.0
        assert s.device_mem_info()[0] < free0
    assert d.ptr is None
    d.free()
    assert s.device_mem_info()[0] >= free0
    s.device_array((1 << 20,))      # dropped on the floor
    gc.collect()
    assert s.device_mem_info()[0] >= free0
    keep = s.device_array((1 << 20,))   # still alive at close(): given back by close()
    pin = s.pinned_empty((1000,))
    pin[:] = np.arange(1000.0)
    view = pin[10:20]
    s.close()
    keep.free()                      # after the handle is gone: a no-op, not a crash
    del pin
    gc.collect()
    assert view.sum() == sum(range(10, 20))   # the block lives as long as a view does
    del view
    gc.collect()
