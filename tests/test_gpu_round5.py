"""GPU parity tests added in round 5 (-m gpu): the HIP path over the part of the reference's public surface no earlier
fixture reached — `ScanSimulator2D(num_beams, fov, eps, theta_dis, max_range)` with every keyword away from its default
(laser_models.py:360-381), yaml origins with a yaw and odd resolutions (laser_models.py:55-86, :417-420), and the
reference's default `F110Env()` (vegas, f110_env.py:104-159).  All three fixtures were recorded by RUNNING the reference
(oracle/refshim/gen_golden.py scan_ctor / scan_rotated / env_defaults).  Nothing here reads /root/reference.
"""
import os

import numpy as np
import pytest

from _util import gold, load_any_map_image, map_stem, rel_err, write_variant_yaml

pytestmark = pytest.mark.gpu
FTOL = 1e-9


@pytest.fixture(scope="module")
def amd():
    import f1tenth_gym_amd
    from f1tenth_gym_amd import _ffi
    assert _ffi.device_count() >= 1, "no MI355X visible: the HIP path cannot run (no CPU fallback)"
    return f1tenth_gym_amd


def _step_scans_at(amd, poses, img, res, origin, layout, reps=1, **ctor):
    """the scans the STEP kernels produce with the cars at rest at `poses` (zero action from rest leaves the pose alone:
    pid(0, 0, 0, 0) = 0): one env per pose, one car per env, noise off"""
    keep = (poses[:, 2] >= 0.0) & (poses[:, 2] <= 2 * np.pi)     # headings update_pose's yaw wrap (base_classes.py:373-380) leaves alone
    assert keep.sum() >= 2
    poses = np.ascontiguousarray(np.tile(poses[keep], (reps, 1)))     # reps > 1: a batch large enough for the longest-first task order
    s = amd.BatchSim(num_envs=len(poses), num_agents=1, map_layout=layout, **ctor)
    s.set_map_image(img, res, origin)
    s.reset(poses)
    s.step(np.zeros((len(poses), 2)))
    o = s.get("scans", "state", "in_collision")
    s.close()
    assert np.array_equal(o["state"][:, [0, 1, 4]], poses) and not o["state"][:, 3].any()
    n = int(keep.sum())
    for r in range(1, reps):
        assert np.array_equal(o["scans"][r * n:(r + 1) * n], o["scans"][:n])
    return o["scans"][:n], keep


# ---------------------------------------------------------------- ScanSimulator2D keywords away from their defaults
@pytest.mark.parametrize("layout", [0, 3])
@pytest.mark.parametrize("case", range(7))
def test_scan_ctor_variants_vs_reference(amd, case, layout):
    g = gold("scan_ctor_variants")
    assert int(g["n_cases"][0]) == 7
    k = case
    beams, fov, eps, theta_dis, max_range = g["c%d_ctor" % k]
    beams, theta_dis = int(beams), int(theta_dis)
    name = str(g["c%d_map" % k][0])
    poses = g["c%d_poses" % k]
    sim = amd.ScanSimulator2D(beams, fov, eps=eps, theta_dis=theta_dis, max_range=max_range, map_layout=layout)
    assert sim.theta_index_increment == g["c%d_theta_index_increment" % k][0]
    assert sim.set_map(map_stem(name) + ".yaml", ".png") is True
    ranges, hits, lk = sim.scan_batch(poses, want_hits=True, want_lookups=True)
    assert np.array_equal(sim.batch.beam_dir_index_batch(poses[:, 2]), g["c%d_dir_idx" % k])
    assert np.array_equal(hits, g["c%d_hit_rc" % k])
    assert np.array_equal(ranges, g["c%d_scans" % k])
    assert np.array_equal(lk, g["c%d_lookups" % k])
    # the reference's one-pose entry point, with and without its rng / std_dev arguments (laser_models.py:429-454)
    for i in (0, len(poses) - 1):
        assert np.array_equal(sim.scan(poses[i], None), g["c%d_scans" % k][i])
    seed, std = g["c%d_noise_seed_std" % k]
    assert np.array_equal(sim.scan(poses[0], np.random.default_rng(int(seed)), std_dev=std), g["c%d_noisy" % k])
    sim.batch.close()
    # ... and through the kernels env.step() runs (k_scan_rays_agent / k_scan_dirs_agent / k_scan_rays)
    img, res, origin = load_any_map_image(name)
    scans, keep = _step_scans_at(amd, poses, img, res, origin, layout, num_beams=beams, fov=fov, eps=eps, theta_dis=theta_dis,
                                 max_range=max_range)
    assert np.array_equal(scans, g["c%d_scans" % k][keep])
    if layout == 3 and case in (0, 1, 2, 6):      # the same through the big-batch form of the scan (12 000+ tasks)
        scans, keep = _step_scans_at(amd, poses, img, res, origin, layout, reps=400, num_beams=beams, fov=fov, eps=eps, theta_dis=theta_dis,
                                     max_range=max_range)
        assert np.array_equal(scans, g["c%d_scans" % k][keep])


# ---------------------------------------------------------------- rotated origin, odd resolution
@pytest.mark.parametrize("layout", [0, 3])
@pytest.mark.parametrize("case", range(3))
def test_scan_rotated_origin_vs_reference(amd, case, layout, tmp_path):
    g = gold("scan_rotated_origin")
    k = case
    name = str(g["r%d_map" % k][0])
    res, origin = float(g["r%d_resolution" % k][0]), [float(v) for v in g["r%d_origin" % k]]
    assert origin[2] != 0.0
    poses = g["r%d_poses" % k]
    yaml_path = write_variant_yaml(tmp_path, name, res, origin)
    sim = amd.ScanSimulator2D(1080, 4.7, map_layout=layout)
    sim.set_map(yaml_path, ".png")
    assert sim.orig_s == np.sin(origin[2]) and sim.orig_c == np.cos(origin[2]) and sim.map_resolution == res
    ranges, hits, lk = sim.scan_batch(poses, want_hits=True, want_lookups=True)
    assert np.array_equal(sim.batch.beam_dir_index_batch(poses[:, 2]), g["r%d_dir_idx" % k])
    assert np.array_equal(hits, g["r%d_hit_rc" % k])
    assert np.array_equal(ranges, g["r%d_scans" % k])
    assert np.array_equal(lk, g["r%d_lookups" % k])
    sim.batch.close()
    img, _, _ = load_any_map_image(name)
    for reps in ((1, 300) if layout == 3 else (1,)):
        scans, keep = _step_scans_at(amd, poses, img, res, origin, layout, reps=reps)
        assert np.array_equal(scans, g["r%d_scans" % k][keep])


@pytest.mark.parametrize("E", [1, 5])
def test_simulator_rollout_on_rotated_origin_vs_reference(amd, E, tmp_path):
    """the reference's 2-car Simulator on berlin under resolution 0.07 and an origin yawed by 0.3 rad: 220 steps, seed-12345
    noise, a wall hit — through the reference-compatible Simulator class (E = 1) and through BatchSim with every env
    replaying it (E = 5)"""
    g = gold("scan_rotated_origin")
    name = str(g["r0_map"][0])
    res, origin = float(g["r0_resolution"][0]), [float(v) for v in g["r0_origin"]]
    yaml_path = write_variant_yaml(tmp_path, name, res, origin)
    acts = g["sim_actions"]
    T = acts.shape[0]
    full = {int(t): g["sim_scans_t%d" % t] for t in g["sim_full_steps"]}
    params = dict(zip(amd._ffi.PARAM_KEYS, g["params"]))
    worst = 0.0
    if E == 1:
        sim = amd.Simulator(params, 2, int(g["sim_seed"][0]))
        sim.set_map(yaml_path, ".png")
        sim.reset(g["sim_start"])
        for t in range(T):
            obs = sim.step(acts[t])
            assert np.array_equal(obs['collisions'], g["sim_collisions"][t]), t
            assert np.array_equal(sim.collision_idx, g["sim_collision_idx"][t]), t
            assert np.array_equal([int(a.in_collision) for a in sim.agents], g["sim_in_collision"][t]), t
            sc = np.array(obs['scans'])
            worst = max(worst, rel_err(np.array([a.state for a in sim.agents]), g["sim_states"][t]),
                        rel_err(sc[:, ::24], g["sim_scans_sub24"][t]), rel_err(sc.sum(axis=1), g["sim_scans_sum"][t]))
            if t in full:
                assert rel_err(sc, full[t]) < FTOL
    else:
        s = amd.BatchSim(params, num_envs=E, num_agents=2)
        s.set_map(yaml_path, ".png")
        s.set_noise_rng(int(g["sim_seed"][0]), 0.01)
        s.reset(np.tile(g["sim_start"], (E, 1)))
        rep = lambda x: np.tile(x, (E,) + (1,) * (x.ndim - 1))
        for t in range(T):
            s.step(np.tile(acts[t], (E, 1)))
            o = s.get("scans", "state", "collisions", "collision_idx", "in_collision")
            assert np.array_equal(o["collisions"], rep(g["sim_collisions"][t])), t
            assert np.array_equal(o["collision_idx"], rep(g["sim_collision_idx"][t])), t
            assert np.array_equal(o["in_collision"], rep(g["sim_in_collision"][t])), t
            worst = max(worst, rel_err(o["state"], rep(g["sim_states"][t])), rel_err(o["scans"][:, ::24], rep(g["sim_scans_sub24"][t])),
                        rel_err(o["scans"].sum(axis=1), rep(g["sim_scans_sum"][t])))
            if t in full:
                assert rel_err(o["scans"], rep(full[t])) < FTOL
        s.close()
    assert worst < FTOL, worst
    assert g["sim_in_collision"].any()


# ---------------------------------------------------------------- F110Env() with no keyword at all
def _check_env_step(g, ep, k, obs, done, info, toggles, near, vec):
    e = lambda name: g["ep%d_%s" % (ep, name)][k]
    pick = (lambda v: np.asarray(v)[0]) if vec else (lambda v: np.asarray(v))
    got = np.stack([pick(obs['poses_x']), pick(obs['poses_y']), pick(obs['poses_theta']), pick(obs['linear_vels_x']), pick(obs['ang_vels_z'])])
    assert rel_err(got, np.stack([e("x"), e("y"), e("th"), e("v"), e("w")])) < FTOL, (ep, k)
    assert np.array_equal(pick(obs['collisions']), e("col")), (ep, k)
    assert np.array_equal(pick(obs['lap_counts']), e("lap_count")), (ep, k)
    assert np.max(np.abs(pick(obs['lap_times']) - e("lap_time"))) < 1e-12
    assert np.array_equal(np.asarray(pick(toggles), dtype=float), e("toggle")), (ep, k)
    assert np.array_equal(np.asarray(pick(near), dtype=bool), e("near")), (ep, k)
    assert np.array_equal(np.asarray(pick(info['checkpoint_done']), dtype=bool), e("ckpt")), (ep, k)
    assert bool(pick(done) if vec else done) == bool(e("done")), (ep, k)
    if 'scans' in obs:
        assert rel_err(np.asarray(pick(obs['scans'])).sum(axis=1), e("scan_sum")) < FTOL, (ep, k)


def test_f110env_without_keywords_vs_reference(amd):
    """`F110Env()`: vegas from inside the package, 2 agents, ego_idx 0, seed 12345, RK4, 0.01 s (f110_env.py:104-159)"""
    g = gold("env_episode_defaults")
    env = amd.F110Env()
    assert env.num_agents == 2 and env.ego_idx == 0 and env.timestep == 0.01 and env.seed == 12345
    assert os.path.basename(env.map_path) == "vegas.yaml"
    for ep in range(2):
        obs, r, done, info = env.reset(g["ep%d_start" % ep])
        assert r == 0.01 and obs['ego_idx'] == 0
        _check_env_step(g, ep, 0, obs, done, info, env.toggle_list, env.near_starts, False)
        for t, a in enumerate(g["ep%d_actions" % ep]):
            obs, r, done, info = env.step(a)
            _check_env_step(g, ep, t + 1, obs, done, info, env.toggle_list, env.near_starts, False)
        assert done


@pytest.mark.parametrize("E", [1, 4])
def test_vec_env_without_keywords_vs_reference(amd, E):
    """the same two episodes through F110VecEnv(E, device_logic=True) with no map / agent keyword: vegas, episode logic on
    the device"""
    g = gold("env_episode_defaults")
    env = amd.F110VecEnv(E, device_logic=True, copy_obs=True)
    for ep in range(2):
        start = np.tile(g["ep%d_start" % ep][None], (E, 1, 1))
        obs, r, done, info = env.reset(start)
        for k in range(len(g["ep%d_actions" % ep]) + 1):
            if k:
                obs, r, done, info = env.step(np.tile(g["ep%d_actions" % ep][k - 1][None], (E, 1, 1)))
            for e in range(E):
                sel = {kk: (v[e:e + 1] if isinstance(v, np.ndarray) else v) for kk, v in obs.items()}
                inf = {kk: v[e:e + 1] for kk, v in info.items()}
                _check_env_step(g, ep, k, sel, done[e:e + 1], inf, inf['toggle_list'], inf['near_starts'], True)
        assert done.all()
    env.sim.batch.close()


# ---------------------------------------------------------------- overlapped gather between two-block steps
def test_overlapped_obs_gather_between_two_block_steps(amd):
    """step, step (two env blocks), overlapped gather WITH the scalar block (k_pack_obs reads state[] on the main stream),
    step (two blocks again) ...: the block gathered after step t must hold step t's poses, not the next step's, for every
    env of BOTH blocks — compared with a handle that always steps as one block (step_groups = 1)"""
    from _util import bench_start_poses, load_map_image
    img, res, origin = load_map_image("example_map")
    E, A, B, T = 8192, 2, 1080, 14           # 16 384 agents: a size the automatic choice runs as two blocks
    N = E * A
    sims = [amd.BatchSim(num_envs=E, num_agents=A, step_groups=g) for g in (0, 1)]
    poses = bench_start_poses(E, A)
    for s in sims:
        s.set_map_image(img, res, origin); s.set_noise_rng(12345, 0.01); s.reset(poses)
        s.comm_init(1, 0, amd.BatchSim.comm_unique_id()); s.comm_set_overlap(True)
    if sims[0].step_groups()[0] < 2:
        pytest.skip("no second stream observed to run concurrently on this box: the handle steps as one block")
    rng = np.random.default_rng(5)
    acts = [np.stack([rng.uniform(-0.3, 0.3, N), rng.uniform(1, 7, N)], axis=1) for _ in range(4)]
    d_act = [[s.device_array((N, 2)) for _ in acts] for s in sims]
    for s, bufs in zip(sims, d_act):
        for b, a in zip(bufs, acts):
            b.upload(a)
    recv = [[(s.device_array((1, N, B)), s.device_array((1, 7, N))) for _ in range(2)] for s in sims]
    blocks = set()
    for cycle in range(T // 4):
        # step (one block: the download below touched the handle), step (two blocks), gather, step (two blocks), gather, step
        for k, s in enumerate(sims):
            s.step_device(d_act[k][0]); s.step_device(d_act[k][1])
            s.comm_all_gather_obs(*recv[k][0])
            s.step_device(d_act[k][2])
            if k == 0:
                blocks.add(s.step_groups()[2])
            s.comm_all_gather_obs(*recv[k][1])
            s.step_device(d_act[k][3])
        for r in range(2):
            assert np.array_equal(recv[0][r][1].download(), recv[1][r][1].download()), (cycle, r)     # the [7][N] scalar block
            assert np.array_equal(recv[0][r][0].download(), recv[1][r][0].download()), (cycle, r)     # the scans
        assert np.array_equal(sims[0].get("state")["state"], sims[1].get("state")["state"])
    assert 2 in blocks
    for s in sims:
        s.comm_set_overlap(False); s.close()


# ---------------------------------------------------------------- lab: the lane-refill scan (survivor compaction)
def test_racecar_class_vs_reference(amd):
    """f110_gym.envs.base_classes.RaceCar (base_classes.py:45-449) driven directly, as the reference's own fixtures were
    recorded: update_pose single steps (RK4, Euler, offset lidar; steer-delay FIFO) and the 400-step rollout of update_pose.npz,
    check_ttc's state zeroing, ray_cast_agents against raycast.npz, update_scan on a list of scans"""
    from f110_gym.envs.base_classes import Integrator, RaceCar
    g = gold("update_pose")
    params = dict(zip(amd._ffi.PARAM_KEYS, g["params"]))
    for name, integ, ld in (("rk4", Integrator.RK4, 0.0), ("euler", Integrator.Euler, 0.0), ("rk4_lidar", Integrator.RK4, 0.275)):
        # class-level state, base_classes.py:64-67: only the FIRST car of a process builds the scan simulator and gets a scan_rng
        # before its first reset (:113-116) — every car of this test is a first car, as in the generator of the fixture
        if RaceCar.scan_simulator is not None:
            RaceCar.scan_simulator.batch.close()
        RaceCar.scan_simulator = None
        car = RaceCar(params, 12345, is_ego=True, time_step=0.01, integrator=integ, lidar_dist=ld)
        car.set_map(map_stem("example_map") + ".yaml", ".png")
        seen = []
        scan_fn = RaceCar.scan_simulator.scan
        RaceCar.scan_simulator.scan = lambda pose, rng, std_dev=0.01: (seen.append(np.array(pose)) or scan_fn(pose, rng, std_dev))
        for i in range(0, len(g[name + "_state0"]), 3):
            car.state = g[name + "_state0"][i].copy()
            car.steer_buffer = g[name + "_buf0"][i, :g[name + "_cnt0"][i]].copy()
            scan = car.update_pose(*g[name + "_action"][i])
            assert scan.shape == (1080,)
            assert rel_err(car.state, g[name + "_state1"][i]) < FTOL
            assert car.steer_buffer.shape[0] == g[name + "_cnt1"][i] and np.array_equal(car.steer_buffer, g[name + "_buf1"][i, :g[name + "_cnt1"][i]])
            assert rel_err(seen[-1], g[name + "_scan_pose"][i]) < FTOL
        if name != "euler":
            car.reset(np.array([0.7, 0.0, 1.37079632679]))
            worst = 0.0
            for t in range(0, 400):
                car.update_pose(*g[name + "_roll_actions"][t])
                worst = max(worst, rel_err(car.state, g[name + "_roll_states"][t]))
            assert worst < 1e-5, worst        # (north_star's rollout bar; measured ~1e-13)
        RaceCar.scan_simulator.scan = scan_fn
    # check_ttc: a wall 5 cm ahead at speed -> collision, state[3:] zeroed (base_classes.py:240-262)
    car.reset(np.array([0.7, 0.0, 1.37079632679])); car.state[3] = 6.0; car.state[5] = 0.3
    scan = np.full(1080, 10.0); scan[540] = RaceCar.side_distances[540] + 0.01
    assert car.check_ttc(scan) is True and car.in_collision and not car.state[3:].any()
    assert car.check_ttc(np.full(1080, 10.0)) is False and not car.in_collision
    # ray_cast_agents / update_scan against the reference's ray_cast outputs
    r = gold("raycast")
    for i in (0, 5, 12, 40, 77):
        car.state[:] = 0.0; car.state[0:2] = r["ego"][i, :2]; car.state[4] = r["ego"][i, 2]
        car.update_opp_poses(r["opp"][i:i + 1])
        scans = [np.full(1080, float(r["base"][0])), np.full(1080, float(r["base"][0]))]
        car.update_scan(scans, 1)
        assert rel_err(scans[1], r["scans"][i]) < FTOL and np.array_equal(scans[0], np.full(1080, float(r["base"][0])))
    RaceCar.scan_simulator.batch.close()
    RaceCar.scan_simulator = None


# ---------------------------------------------------------------- RL-loop hand-off: DLPack, the on-device scan consumer
def _scan_policy_numpy(scans, fov, steer_gain=0.5, steer_max=0.4189, sector_limit=1.75, v_lo=1.0, v_hi=6.0, d_ref=6.0):
    """f110_scan_policy_device restated (include/f110.h): same sector bounds, same summation order"""
    n, B = scans.shape
    out = np.empty((n, 2))
    inc = fov / (B - 1)
    for a in range(n):
        best, best_c, front = -np.inf, 0.0, np.inf
        for s in range(64):
            b0, b1 = (s * B + 63) // 64, ((s + 1) * B + 63) // 64
            seg = scans[a, b0:b1]
            tot = 0.0
            for r in seg:
                tot += r
            centre = -fov / 2. + inc * (0.5 * (b0 + b1 - 1))
            if b1 > b0 and abs(centre) <= sector_limit and tot / (b1 - b0) > best:
                best, best_c = tot / (b1 - b0), centre
            if 28 <= s < 36:
                front = min(front, seg.min())
        f = front / d_ref
        out[a] = [min(max(steer_gain * best_c, -steer_max), steer_max), v_lo + (v_hi - v_lo) * min(f, 1.0)]
    return out


def test_scan_policy_device_reads_the_scans_in_place(amd):
    from _util import bench_start_poses, load_map_image
    E, A = 24, 2
    s = amd.BatchSim(num_envs=E, num_agents=A)
    s.set_map_image(*load_map_image("example_map")); s.set_noise_rng(12345, 0.01)
    s.reset(bench_start_poses(E, A))
    act = s.device_array((E * A, 2))
    act.upload(np.zeros((E * A, 2)))
    for t in range(6):
        s.step_device(act)
        s.scan_policy_device(act)
        want = _scan_policy_numpy(s.get("scans")["scans"], 4.7)
        assert np.array_equal(act.download(), want), t
    assert np.ptp(want[:, 0]) > 0 and np.ptp(want[:, 1]) > 0
    s.close()


def test_dlpack_hand_off_to_torch():
    """DeviceArray.__dlpack__ / __dlpack_device__: torch wraps the simulator's scan buffer and the action buffer without a copy
    (kDLROCM), sees the step's values, and what it writes into the action buffer is what the next step integrates.  In a process
    of its own: torch has to be imported BEFORE this package's library so that both share one HIP runtime."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "dlpack_torch_worker.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=600)
    if "SKIP" in out.stdout:
        pytest.skip(out.stdout.strip().splitlines()[-1])
    assert out.returncode == 0 and "DLPACK OK" in out.stdout, (out.stdout[-800:], out.stderr[-1500:])


def test_example_rl_loop_device_runs():
    """examples/rl_loop_device.py end to end: the built-in scan-consuming policy, and (where torch sees the GPU) the torch MLP fed
    through DLPack on the simulator's own stream"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ex = os.path.join(root, "examples", "rl_loop_device.py")
    out = subprocess.run([sys.executable, ex, "--envs", "256", "--steps", "60"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0 and "agent-steps/s" in out.stdout and "built-in scan" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])
    probe = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.is_available())"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    if probe.returncode != 0 or "True" not in probe.stdout:
        return      # no torch / torch without a GPU here: the DLPack leg of the example cannot run
    out = subprocess.run([sys.executable, ex, "--envs", "256", "--steps", "40", "--torch"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0 and "torch MLP via DLPack" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])


NESTED = bool(os.environ.get("F110_NESTED_SUITE"))   # the lab build's re-run of the suite runs a few seeds of every chunk


def _fuzzer(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "debug", name + ".py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    return fz


@pytest.mark.parametrize("first", range(0, 200, 50))
def test_fuzz_envs_bounded_seeds(amd, first):
    """tools/debug/fuzz_envs.py, seeds 0 .. 199 in the driver-run suite (round 6; 8 before, 850 by hand in round 5): a different track
    per env (f110_add_map_dt / f110_set_env_maps), a vehicle parameter set per agent or per slot, constructor arguments and yawed origins
    drawn together, calm actions so that the rollouts run their length through wall hits and car-to-car hits — the HIP step against
    one CPU oracle per env; ~0.3 s per seed on the box"""
    fz = _fuzzer("fuzz_envs")
    bad = [sd for sd in range(first, first + (2 if NESTED else 50)) if not fz.run(sd)]
    assert not bad, bad


@pytest.mark.parametrize("first", range(0, 300, 100))
def test_fuzz_episode_bounded_seeds(amd, first):
    """tools/debug/fuzz_episode.py, seeds 0 .. 299 in the driver-run suite (round 6; 10 before, 2 200 by hand in round 5):
    F110VecEnv(device_logic=True) — lap toggles, counts, times, done and the auto-reset re-seats done by the finalize kernels
    (f110_env.py:219-306) — equals the host-side bookkeeping (pinned to the live reference by tests/test_reference_fuzz.py) over random
    tracks, 1-4 cars, ego indices, time steps, integrators, partial resets, with half of the envs driven in circles so that laps
    complete; ~0.1 s per seed on the box"""
    fz = _fuzzer("fuzz_episode")
    bad = [sd for sd in range(first, first + (3 if NESTED else 100)) if not fz.run(sd)]
    assert not bad, bad
