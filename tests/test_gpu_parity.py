"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the oracle and the
golden vectors.  Bars (BASELINE.json north_star): collision / wall flags and terminating ray cells
bit-exact; scan ranges and vehicle state within 1e-5 relative (measured: ~1e-15, asserted at 1e-9
for unit inputs).  Nothing here reads /root/reference.
"""
import numpy as np
import pytest

from _util import gold, load_map_image, oracle_map_dt, bench_start_poses, raceline, rel_err

pytestmark = pytest.mark.gpu

FTOL = 1e-9        # unit-level float tolerance (device libm vs glibc: a few ulp)
NORTH_STAR = 1e-5  # rollout-level bar from BASELINE.json


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


# ---------------------------------------------------------------------------- map / EDT (a9)
@pytest.mark.parametrize("name", ["example_map", "berlin", "skirk"])
def test_device_map_pipeline_bit_exact(amd, name):
    img, res, origin = load_map_image(name)
    dt, _, _ = oracle_map_dt(name)
    s = amd.BatchSim(num_envs=1, num_agents=1)
    s.set_map_image(img, res, origin)
    assert np.array_equal(s.get_map_dt(), dt)
    s.close()


def test_device_edt_random_images(unit, orc):
    rng = np.random.default_rng(5)
    for shape, p in [((37, 53), 0.02), ((64, 64), 0.3), ((5, 200), 0.01), ((120, 7), 0.1), ((1, 40), 0.2), ((300, 257), 0.001)]:
        img = (rng.random(shape) > p).astype(np.uint8) * 255
        img[rng.integers(shape[0]), rng.integers(shape[1])] = 0
        assert np.array_equal(unit.edt_sq(img), orc.edt_sq(img))


# ---------------------------------------------------------------------------- dynamics (a1-a7)
def test_dynamics_and_pid(unit, orc):
    g = gold("dynamics")
    f_st, f_ks = unit.dynamics_batch(g["x"], g["u"], g["params"])
    ref_st = np.array([orc.vehicle_dynamics_st(x, u, g["params"]) for x, u in zip(g["x"], g["u"])])
    assert rel_err(f_st, ref_st) < FTOL and rel_err(f_st, g["f_st"]) < FTOL
    assert rel_err(f_ks, g["f_ks"]) < FTOL
    assert np.array_equal(unit.pid_batch(g["pid_in"], g["params"]), g["pid_out"])
    # reference known answer, dynamic_models.py:255-279
    p = orc.params_vec({'mu': 1.0489, 'C_Sf': 21.92 / 1.0489, 'C_Sr': 21.92 / 1.0489, 'lf': 0.3048 * 3.793293,
                        'lr': 0.3048 * 4.667707, 'h': 0.3048 * 2.01355, 'm': 4.4482216152605 / 0.3048 * 74.91452,
                        'I': 4.4482216152605 * 0.3048 * 1321.416, 's_min': -1.066, 's_max': 1.066, 'sv_min': -0.4,
                        'sv_max': 0.4, 'v_min': -13.6, 'v_max': 50.8, 'v_switch': 7.319, 'a_max': 11.5})
    x_st = np.array([[2.0233348142065677, 0.0041907137716636, 0.0197545248559617, 15.7216236334290116,
                      0.0025857914776859, 0.0529001056654038, 0.0033012170610298]])
    x_ks = np.zeros((1, 7)); x_ks[0, :5] = [3.9579422297936526, 0.0391650102771405, 0.0378491427211811, 16.3546957860883566, 0.0294717351052816]
    u = np.array([[0.15, 0.63 * 9.81]])
    f_st, _ = unit.dynamics_batch(x_st, u, p)
    _, f_ks = unit.dynamics_batch(x_ks, u, p)
    assert np.max(np.abs(f_st[0] - [15.7213512030862397, 0.0925527979719355, 0.15, 5.3536773276413925, 0.0529001056654038, 0.6435589397748606, 0.0313297971641291])) < 1e-7
    assert np.max(np.abs(f_ks[0] - [16.3475935934250209, 0.4819314886013121, 0.15, 5.1464424102339752, 0.2401426578627629])) < 1e-7


@pytest.mark.parametrize("name,integ,ld", [("rk4", 1, 0.0), ("euler", 2, 0.0), ("rk4_lidar", 1, 0.275)])
def test_update_pose(unit, name, integ, ld):
    g = gold("update_pose")
    s1, b1, c1, sp = unit.update_pose_batch(g[name + "_state0"], g[name + "_buf0"], g[name + "_cnt0"], g[name + "_action"],
                                            g["params"], 0.01, integ, ld)
    assert rel_err(s1, g[name + "_state1"]) < FTOL
    assert rel_err(sp, g[name + "_scan_pose"]) < FTOL
    assert np.array_equal(c1, g[name + "_cnt1"])
    for i, c in enumerate(c1):
        assert np.array_equal(b1[i, :c], g[name + "_buf1"][i, :c])
    if name != "euler":   # 400-step rollout from reset through the step-by-step entry point
        st = np.zeros((1, 7)); st[0, 0], st[0, 1], st[0, 4] = 0.7, 0.0, 1.37079632679
        sb = np.zeros((1, 2)); cnt = np.zeros(1, dtype=np.int32); worst = 0.0
        for t, a in enumerate(g[name + "_roll_actions"]):
            st, sb, cnt, _ = unit.update_pose_batch(st, sb, cnt, a.reshape(1, 2), g["params"], 0.01, integ, ld)
            worst = max(worst, rel_err(st[0], g[name + "_roll_states"][t]))
        assert worst < FTOL, worst


def test_bad_integrator_raises(amd, unit):
    with pytest.raises(SyntaxError):
        unit.update_pose_batch(np.zeros((1, 7)), np.zeros((1, 2)), np.zeros(1, dtype=np.int32), np.zeros((1, 2)),
                               amd.DEFAULT_PARAMS, 0.01, 7, 0.0)
    with pytest.raises(SyntaxError):
        amd.Simulator(amd.DEFAULT_PARAMS, 1, 1, integrator="Heun")


# ---------------------------------------------------------------------------- scan (a8-a12)
@pytest.mark.parametrize("layout", [0, 3])
@pytest.mark.parametrize("fixture,mapname,beams,fov", [
    ("scan_example_map", "example_map", 1080, 4.7), ("scan_berlin", "berlin", 1080, 4.7),
    ("scan_example_map_4096", "example_map", 4096, 4.7), ("scan_example_map_271", "example_map", 271, 6.0)])
def test_scan_bit_exact_vs_golden(amd, fixture, mapname, beams, fov, layout):
    g = gold(fixture)
    img, res, origin = load_map_image(mapname)
    s = amd.BatchSim(num_envs=1, num_agents=1, num_beams=beams, fov=fov, map_layout=layout)
    s.set_map_image(img, res, origin)
    ranges, hits, lk = s.scan_batch(g["poses"], want_hits=True, want_lookups=True)
    assert np.array_equal(s.beam_dir_index_batch(g["poses"][:, 2]), g["dir_idx"])
    assert np.array_equal(hits, g["hit_rc"])        # terminating cells: bit-exact
    assert np.array_equal(ranges, g["scans"])       # same table + same op order: bit-exact
    assert np.array_equal(lk, g["lookups"])
    s.close()


def test_scan_generic_paths_vs_oracle(amd, orc):
    """non-power-of-two resolution and rotated origins (guarded division / rotation code)"""
    dt, res, origin = oracle_map_dt("example_map")
    sub = np.ascontiguousarray(dt[600:1000, 900:1300])
    rng = np.random.default_rng(11)
    for layout in (0, 3):
        for res2, org in [(0.05, [-3.0, -4.0, 0.3]), (0.0625, [1.0, 2.0, -1.1]), (0.07, [0.0, 0.0, 0.0])]:
            so = orc.ScanOracle(1080, 4.7)
            so.set_map_dt(sub * (res2 / res), res2, org)
            s = amd.BatchSim(num_envs=1, num_agents=1, map_layout=layout)
            s.set_map_dt(so.dt, res2, org)
            c, sn = np.cos(org[2]), np.sin(org[2])
            uv = rng.uniform(5, 15, (24, 2)) * res2 / 0.05
            poses = np.stack([org[0] + c * uv[:, 0] - sn * uv[:, 1], org[1] + sn * uv[:, 0] + c * uv[:, 1],
                              rng.uniform(0, 6.28, 24)], axis=1)
            ranges, hits = s.scan_batch(poses, want_hits=True)
            for k in range(24):
                ref, ref_hits = so.scan(poses[k], want_hits=True)
                assert np.array_equal(hits[k], ref_hits) and np.array_equal(ranges[k], ref)
            s.close()


def test_scan_before_map_raises(amd):
    s = amd.ScanSimulator2D(1080, 4.7)
    with pytest.raises(ValueError):
        s.scan(np.zeros(3), None)
    s.batch.close()
    b = amd.BatchSim(num_envs=1, num_agents=1)
    with pytest.raises(ValueError):
        b.step(np.zeros((1, 2)))
    b.close()


def test_scan_simulator_class(amd):
    """ScanSimulator2D API (laser_models.py:348-457) incl. ScanTests.test_rng semantics :554-580"""
    import os
    from _util import MAPS
    s = amd.ScanSimulator2D(1080, 4.7)
    assert s.set_map(os.path.join(MAPS, "berlin.yaml"), ".png") is True
    pose = np.array([0., 0., 0.])
    a = s.scan(pose, np.random.default_rng(seed=12345))
    b = s.scan(pose, np.random.default_rng(seed=12345))
    rng = np.random.default_rng(seed=12345)
    c1 = s.scan(pose, rng); c2 = s.scan(pose, rng)
    assert np.array_equal(a, b) and np.array_equal(a, c1) and not np.array_equal(c1, c2)
    clean = s.scan(pose, None)
    assert np.allclose(a - clean, np.random.default_rng(seed=12345).normal(0., 0.01, 1080), atol=1e-15)
    assert s.get_increment() == 4.7 / 1079
    legacy = gold("legacy_scan")["berlin"]          # unittest/scan_sim.py:321-342: MSE < 2
    scans = s.scan_batch(np.array([[0., 0., th] for th in np.linspace(-1., 1., 10)]))
    assert np.mean((scans - legacy) ** 2) < 2.0
    s.batch.close()


# ---------------------------------------------------------------------------- ttc / collision / raycast
def test_ttc(unit):
    g = gold("ttc")
    assert np.array_equal(unit.scan_angles, g["scan_angles"])
    assert np.array_equal(unit.cosines, g["cosines"]) and np.array_equal(unit.side_distances, g["side_distances"])
    assert np.array_equal(unit.ttc_batch(g["scans"], g["vels"], 0.005), g["flags"])


def test_vertices_gjk_collision_multiple(unit, orc):
    g = gold("collision")
    L, W = g["length"][0], g["width"][0]
    assert rel_err(unit.get_vertices_batch(g["pose_a"], L, W), g["vert_a"]) < FTOL
    assert np.array_equal(unit.gjk_batch(g["vert_a"], g["vert_b"]), g["flags"])
    allv = np.array([[orc.get_vertices(p, L, W) for p in gp] for gp in g["group_poses"]])
    col, idx = unit.collision_multiple_batch(allv)
    assert np.array_equal(col, g["group_collisions"]) and np.array_equal(idx, g["group_idx"])
    # reference known answers, collision_models.py:306-324
    np.random.seed(1234)
    v1 = np.asarray([[4, 11.], [5, 5], [9, 9], [10, 10]])
    a = np.array([v1 + np.random.normal(size=v1.shape) / 100. for _ in range(2000)])
    assert np.all(unit.gjk_batch(a[0::2], a[1::2]) == 1)
    np.random.seed(1234)
    bodies = [v1 + np.random.normal(size=v1.shape) / 100. for _ in range(6)] + [v1 + 10.]
    col, idx = unit.collision_multiple_batch(np.stack(bodies)[None])
    assert np.array_equal(col[0], [1., 1., 1., 1., 1., 1., 0.]) and np.array_equal(idx[0], [5., 5., 5., 5., 5., 4., -1.])


def test_raycast_and_get_range(unit):
    g = gold("raycast")
    n = g["ego"].shape[0]
    out, mm = unit.raycast_batch(g["ego"], g["vertices"], np.full((n, 1080), g["base"][0]))
    assert np.array_equal(mm[:, 0], g["min_ind"]) and np.array_equal(mm[:, 1], g["max_ind"])   # window: exact
    ref = g["scans"]
    assert np.array_equal(out != g["base"][0], ref != g["base"][0])                            # same beams touched
    assert rel_err(out, ref) < FTOL
    gr = unit.get_range_batch(g["get_range_in"])
    assert np.array_equal(np.isinf(gr), np.isinf(g["get_range_out"]))
    fin = ~np.isinf(gr)
    assert rel_err(gr[fin], g["get_range_out"][fin]) < FTOL


def test_free_functions(amd):
    """envs/__init__.py star-exports, original signatures"""
    g = gold("dynamics"); p = g["params"]
    f = amd.vehicle_dynamics_st(g["x"][9], g["u"][9], *p[:16])
    assert rel_err(f, g["f_st"][9]) < FTOL
    f = amd.vehicle_dynamics_ks(g["x"][9][:5], g["u"][9], *p[:16])
    assert rel_err(f, g["f_ks"][9]) < FTOL
    r = g["pid_in"][5]
    assert amd.pid(r[0], r[1], r[2], r[3], p[11], p[13], p[15], p[14]) == tuple(g["pid_out"][5])
    c = gold("collision")
    assert rel_err(amd.get_vertices(c["pose_a"][3], 0.58, 0.31), c["vert_a"][3]) < FTOL
    assert amd.collision(c["vert_a"][3], c["vert_b"][3]) == bool(c["flags"][3])
    t = gold("ttc")
    assert amd.check_ttc_jit(t["scans"][1], t["vels"][1], t["scan_angles"], t["cosines"], t["side_distances"], 0.005) == bool(t["flags"][1])
    rc = gold("raycast")
    scan = np.full(1080, rc["base"][0])
    out = amd.ray_cast(rc["ego"][40], scan, rc["scan_angles"], rc["vertices"][40])
    assert out is scan and rel_err(scan, rc["scans"][40]) < FTOL


# ---------------------------------------------------------------------------- Simulator.step (a20)
def _noise(T, B=1080, seed=12345):
    return np.random.default_rng(seed).normal(0., 0.01, size=(T, B))


@pytest.mark.parametrize("layout", [0, 3])
def test_sim_rollout_vs_golden(amd, layout):
    g = gold("sim_rollout")
    img, res, origin = load_map_image("example_map")
    T = g["actions"].shape[0]
    s = amd.BatchSim(dict(zip(amd._ffi.PARAM_KEYS, g["params"])), num_envs=1, num_agents=2, map_layout=layout)
    s.set_map_image(img, res, origin)
    s.set_noise_table(_noise(T))
    s.reset(g["start"])
    full = {int(t): g["scans_t%d" % t] for t in g["full_steps"]}
    worst_state = worst_scan = 0.0
    for t in range(T):
        s.step(g["actions"][t])
        o = s.get("scans", "state", "collisions", "collision_idx", "in_collision", "agent_poses")
        assert np.array_equal(o["collisions"], g["collisions"][t]), t
        assert np.array_equal(o["in_collision"], g["in_collision"][t]), t
        assert np.array_equal(o["collision_idx"], g["collision_idx"][t]), t
        worst_state = max(worst_state, rel_err(o["state"], g["states"][t]))
        worst_scan = max(worst_scan, rel_err(o["scans"][:, ::8], g["scans_sub8"][t]))
        assert rel_err(o["agent_poses"], g["agent_poses"][t]) < FTOL
        if t in full:
            assert rel_err(o["scans"], full[t]) < FTOL
    assert worst_state < FTOL and worst_scan < FTOL, (worst_state, worst_scan)
    s.close()


def test_simulator_class_and_env_episode(amd):
    """reference-compatible Simulator / F110Env against the golden 2-agent rollout and the
    1-agent lap episode (reset-steps-once, lap toggles, done)"""
    import os
    from _util import MAPS
    g = gold("sim_rollout")
    sim = amd.Simulator(dict(zip(amd._ffi.PARAM_KEYS, g["params"])), 2, 12345)
    sim.set_map(os.path.join(MAPS, "example_map.yaml"), ".png")
    with pytest.raises(ValueError):
        sim.reset(np.zeros((3, 3)))
    with pytest.raises(IndexError):
        sim.update_params(amd.DEFAULT_PARAMS, agent_idx=2)
    sim.reset(g["start"])
    for t in range(60):
        obs = sim.step(g["actions"][t])
        assert np.array_equal(obs['collisions'], g["collisions"][t])
        assert rel_err(np.array([a.state for a in sim.agents]), g["states"][t]) < FTOL
        assert rel_err(np.array(obs['scans'])[:, ::8], g["scans_sub8"][t]) < FTOL
        assert obs['linear_vels_y'] == [0., 0.] and len(obs['scans']) == 2
    e = gold("env_episode")
    env = amd.F110Env(map=os.path.join(MAPS, "example_map"), map_ext='.png', num_agents=1, seed=12345)
    obs, r, done, info = env.reset(e["start"])
    rec = lambda: (obs['poses_x'][0], obs['poses_y'][0], obs['poses_theta'][0], obs['linear_vels_x'][0])
    assert r == 0.01 and rel_err(rec(), (e["x"][0], e["y"][0], e["th"][0], e["v"][0])) < FTOL
    assert rel_err(np.sum(obs['scans'][0]), e["scan_sum"][0]) < FTOL
    for t, a in enumerate(e["actions"]):
        obs, r, done, info = env.step(a.reshape(1, 2))
        k = t + 1
        assert rel_err(rec(), (e["x"][k], e["y"][k], e["th"][k], e["v"][k])) < FTOL
        assert float(obs['lap_counts'][0]) == e["lap_count"][k] and done == bool(e["done"][k])
        assert float(env.toggle_list[0]) == e["toggle"][k] and bool(env.near_starts[0]) == bool(e["near"][k])
        assert abs(float(obs['lap_times'][0]) - e["lap_time"][k]) < 1e-12
        assert float(obs['collisions'][0]) == e["col"][k]
        assert rel_err(np.sum(obs['scans'][0]), e["scan_sum"][k]) < FTOL
    assert done and bool(info['checkpoint_done'][0])


def _drive(amd, orc, E, A, T, layout=0, seed=0, beams=1080, reset_every=None, check_every=1):
    """step the HIP sim and the oracle side by side on the bench inputs"""
    img, res, origin = load_map_image("example_map")
    dt, _, _ = oracle_map_dt("example_map")
    noise = _noise(T + 1, beams)
    s = amd.BatchSim(num_envs=E, num_agents=A, num_beams=beams, map_layout=layout)
    s.set_map_image(img, res, origin); s.set_noise_table(noise)
    ref = orc.SimOracle(E, A, num_beams=beams)
    ref.set_map_dt(dt, res, origin); ref.set_noise(noise)
    poses = bench_start_poses(E, A)
    s.reset(poses); ref.reset(poses)
    rng = np.random.default_rng(seed)
    act = np.zeros((E * A, 2))
    stats = {"state": 0.0, "scan": 0.0, "flag_mismatch": 0, "wall": 0, "gjk": 0}
    for t in range(T):
        if t % 20 == 0:
            act = np.stack([rng.uniform(-0.2, 0.2, E * A), rng.uniform(2.0, 6.0, E * A)], axis=1)
        s.step(act); ref.step(act, 8)
        if reset_every and t % reset_every == reset_every - 1:
            mask = (ref.collisions.reshape(E, A)[:, 0] != 0).astype(np.uint8)
            s.reset(poses, mask); ref.reset(poses, mask)
        if t % check_every == 0 or t == T - 1:
            o = s.get("scans", "state", "collisions", "in_collision", "step_count")
            stats["flag_mismatch"] += int(np.sum(o["collisions"] != ref.collisions) + np.sum(o["in_collision"] != ref.in_collision))
            stats["state"] = max(stats["state"], rel_err(o["state"], ref.state))
            stats["scan"] = max(stats["scan"], rel_err(o["scans"], ref.scans))
            stats["wall"] += int(ref.in_collision.sum()); stats["gjk"] += int(ref.collisions.sum())
            assert np.array_equal(o["step_count"], ref.step_count)
    s.close()
    return stats


@pytest.mark.parametrize("layout", [0, 3])
def test_step_vs_oracle_64_envs_200_steps(amd, orc, layout):
    """the parity gate that accompanies every timing: first 64 envs x 200 steps"""
    st = _drive(amd, orc, 64, 2, 200, layout=layout, reset_every=10, check_every=5)
    assert st["flag_mismatch"] == 0, st
    assert st["state"] < NORTH_STAR and st["scan"] < NORTH_STAR, st
    assert st["wall"] > 0, "rollout never exercised a wall hit"


def test_step_4096_beams_dedupe_vs_oracle(amd, orc):
    """more beams than table directions: the step marches each distinct direction once and
    writes the beams that share it (k_scan_dirs_agent; row-major table: every beam is marched); must equal the oracle"""
    st = _drive(amd, orc, 10, 2, 40, beams=4096, reset_every=8)
    assert st["flag_mismatch"] == 0 and st["state"] < NORTH_STAR and st["scan"] < 1e-12, st
    st = _drive(amd, orc, 7, 1, 25, beams=2500, layout=3)
    assert st["flag_mismatch"] == 0 and st["scan"] < 1e-12, st


def test_step_many_agents_per_env(amd, orc):
    st = _drive(amd, orc, 6, 5, 60)
    assert st["flag_mismatch"] == 0 and st["state"] < NORTH_STAR and st["scan"] < NORTH_STAR, st
    # single-agent envs, long enough for wall hits AND the steps after them (collisions must drop
    # back to 0: collision_multiple returns zeros for one body, collision_models.py:196-197)
    st = _drive(amd, orc, 24, 1, 220)
    assert st["flag_mismatch"] == 0 and st["state"] < NORTH_STAR and st["scan"] < NORTH_STAR, st
    assert st["wall"] > 0


def test_config5_4096_beams_tiled_big_map(amd, orc):
    """BASELINE config 5 shape: 4096 beams on example_map tiled 2x2 (3200x3200 table)"""
    img, res, origin = load_map_image("example_map")
    dt, _, _ = oracle_map_dt("example_map")
    big_img = np.tile(img, (2, 2))
    big_dt = np.tile(dt, (2, 2))        # NOT the EDT of the tiled image; used via set_map_dt on both sides
    so = orc.ScanOracle(4096, 4.7); so.set_map_dt(big_dt, res, origin)
    s = amd.BatchSim(num_envs=1, num_agents=1, num_beams=4096, map_layout=0)
    s.set_map_dt(big_dt, res, origin)
    poses = bench_start_poses(12, 1)
    ranges, hits = s.scan_batch(poses, want_hits=True)
    for k in range(12):
        ref, rh = so.scan(poses[k], want_hits=True)
        assert np.array_equal(ranges[k], ref) and np.array_equal(hits[k], rh)
    # and the device EDT of the genuinely tiled picture equals the oracle's
    s.set_map_image(big_img, res, origin)
    assert np.array_equal(s.get_map_dt(), orc.map_dt_from_image(big_img, res))
    s.close()


# ---------------------------------------------------------------------------- full-size properties
@pytest.mark.parametrize("N", [4096, 65536])
def test_full_size_properties(amd, orc, N):
    """BASELINE configs 2/3 at full size, through size-independent properties:
    (1) envs that share start pose + actions produce identical rows (the kernels are
        deterministic and index-independent), (2) a slice of 32 envs matches the oracle,
    (3) ranges stay in [-6 sigma, max_range + 6 sigma], (4) masked reset only touches masked envs."""
    E, A, T = N // 2, 2, 6
    img, res, origin = load_map_image("example_map")
    dt, _, _ = oracle_map_dt("example_map")
    noise = _noise(T + 1)
    s = amd.BatchSim(num_envs=E, num_agents=A)
    s.set_map_image(img, res, origin); s.set_noise_table(noise)
    base = bench_start_poses(783, A).reshape(783, A, 3)
    poses = base[np.arange(E) % 783].reshape(E * A, 3)       # env e and e+783 are twins
    s.reset(poses)
    rng = np.random.default_rng(1)
    a783 = np.stack([rng.uniform(-0.2, 0.2, (783, A)), rng.uniform(2, 6, (783, A))], axis=2)
    act = a783[np.arange(E) % 783].reshape(E * A, 2)
    ref = orc.SimOracle(32, A); ref.set_map_dt(dt, res, origin); ref.set_noise(noise); ref.reset(poses[:64])
    for t in range(T):
        s.step(act); ref.step(act[:64], 8)
    o = s.get("scans", "state", "collisions", "in_collision")
    sc = o["scans"].reshape(E, A, 1080); stt = o["state"].reshape(E, A, 7)
    twins = np.arange(783, min(E, 783 * 3))
    assert np.array_equal(sc[twins], sc[twins % 783]) and np.array_equal(stt[twins], stt[twins % 783])
    assert np.array_equal(o["collisions"][:64], ref.collisions) and np.array_equal(o["in_collision"][:64], ref.in_collision)
    assert rel_err(o["state"][:64], ref.state) < NORTH_STAR and rel_err(o["scans"][:64], ref.scans) < NORTH_STAR
    assert o["scans"].min() > -0.06 and o["scans"].max() < 30.06
    before = s.get("state", "step_count")
    mask = np.zeros(E, dtype=np.uint8); mask[::3] = 1
    s.reset(poses, mask)
    after = s.get("state", "step_count")
    keep = np.repeat(mask == 0, A)
    assert np.array_equal(after["state"][keep], before["state"][keep])
    assert np.all(after["step_count"][~keep] == 0) and np.all(after["step_count"][keep] == T)
    assert np.array_equal(after["state"][~keep][:, [0, 1, 4]], poses[~keep])
    s.close()


def test_device_resident_step_and_views(amd):
    """f110_step_device / device views: the RL-loop form where actions and observations stay in HBM"""
    img, res, origin = load_map_image("example_map")
    E, A = 16, 2
    poses = bench_start_poses(E, A)
    act = np.stack([np.full(E * A, 0.05), np.full(E * A, 3.0)], axis=1)
    a = amd.BatchSim(num_envs=E, num_agents=A); b = amd.BatchSim(num_envs=E, num_agents=A)
    for s in (a, b):
        s.set_map_image(img, res, origin); s.reset(poses)
    d_act = b.device_array((E * A, 2)); d_act.upload(act)
    for _ in range(5):
        a.step(act); b.step_device(d_act)
    b.sync()
    v = b.device_views()
    assert np.array_equal(v["scans"].download(), a.get("scans")["scans"])
    assert np.array_equal(v["state"].download().T, a.get("state")["state"])
    assert v["scans"].__cuda_array_interface__["shape"] == (E * A, 1080)
    d_act.free(); a.close(); b.close()


def test_vec_env_auto_reset(amd):
    import os
    from _util import MAPS
    E = 8
    env = amd.F110VecEnv(E, auto_reset=True, map=os.path.join(MAPS, "example_map"), map_ext='.png', num_agents=2)
    poses = bench_start_poses(E, 2).reshape(E, 2, 3)
    obs, r, done, info = env.reset(poses)
    assert obs['scans'].shape == (E, 2, 1080) and done.shape == (E,)
    act = np.zeros((E, 2, 2)); act[:, :, 0] = 0.4; act[:, 0, 1] = 6.0
    seen_done = False
    for _ in range(150):
        obs, r, done, info = env.step(act)
        if done.any():
            seen_done = True
            st = env.sim.batch.get("state", "step_count")
            idx = np.nonzero(done)[0][0]
            assert st["step_count"][idx * 2] == 0 and np.array_equal(st["state"][idx * 2][[0, 1, 4]], poses[idx, 0])
            break
    assert seen_done


@pytest.mark.parametrize("layout,tasks,block", [(0, 1, 64), (0, 3, 128), (0, 2, 256), (3, 4, 64), (3, 1, 256), (3, 2, 128)])
def test_scan_launch_geometries_bit_exact(amd, orc, layout, tasks, block):
    """map layout / tasks-per-wave / workgroup size only change scheduling and storage:
    unit scans equal the golden vectors and a stepped batch equals the oracle bit-for-bit on scans"""
    g = gold("scan_example_map")
    img, res, origin = load_map_image("example_map")
    s = amd.BatchSim(num_envs=1, num_agents=1, map_layout=layout, scan_tasks_per_wave=tasks, scan_block=block)
    s.set_map_image(img, res, origin)
    ranges, hits, lk = s.scan_batch(g["poses"], want_hits=True, want_lookups=True)
    assert np.array_equal(hits, g["hit_rc"]) and np.array_equal(ranges, g["scans"]) and np.array_equal(lk, g["lookups"])
    s.close()
    dt, _, _ = oracle_map_dt("example_map")
    E, A, T = 37, 2, 12
    noise = _noise(T + 1)
    s = amd.BatchSim(num_envs=E, num_agents=A, map_layout=layout, scan_tasks_per_wave=tasks, scan_block=block)
    s.set_map_image(img, res, origin); s.set_noise_table(noise)
    ref = orc.SimOracle(E, A); ref.set_map_dt(dt, res, origin); ref.set_noise(noise)
    poses = bench_start_poses(E, A); s.reset(poses); ref.reset(poses)
    rng = np.random.default_rng(4)
    for t in range(T):
        act = np.stack([rng.uniform(-0.3, 0.3, E * A), rng.uniform(1.0, 7.0, E * A)], axis=1)
        s.step(act); ref.step(act, 8)
    o = s.get("scans", "state", "collisions", "in_collision")
    assert np.array_equal(o["collisions"], ref.collisions) and np.array_equal(o["in_collision"], ref.in_collision)
    assert rel_err(o["state"], ref.state) < FTOL and rel_err(o["scans"], ref.scans) < FTOL
    s.close()


def test_device_episode_logic_matches_host_and_golden(amd):
    """(f)-1: lap / done bookkeeping and auto-reset on the device == the host _LapLogic (which is
    pinned to the reference episode in test_host_logic.py) and == the golden F110Env episode"""
    import os
    from _util import MAPS
    kw = dict(map=os.path.join(MAPS, "example_map"), map_ext='.png', num_agents=2, seed=12345)
    E = 24
    host = amd.F110VecEnv(E, auto_reset=True, **kw)
    dev = amd.F110VecEnv(E, auto_reset=True, device_logic=True, **kw)
    poses = bench_start_poses(E, 2).reshape(E, 2, 3)
    oh, _, dh, ih = host.reset(poses); od, _, dd, idv = dev.reset(poses)
    rng = np.random.default_rng(9)
    n_done = 0
    for t in range(160):
        if t % 20 == 0:
            act = np.stack([rng.uniform(-0.3, 0.3, (E, 2)), rng.uniform(-1.0, 6.0, (E, 2))], axis=2)
        oh, _, dh, ih = host.step(act); od, _, dd, idv = dev.step(act)
        assert np.array_equal(dh, dd), t
        for k in ("poses_x", "poses_y", "poses_theta", "linear_vels_x", "collisions", "lap_counts", "scans"):
            assert np.array_equal(oh[k], od[k]), (t, k)
        assert np.allclose(oh["lap_times"], od["lap_times"], rtol=0, atol=1e-12)
        assert np.array_equal(ih["checkpoint_done"], idv["checkpoint_done"])
        assert np.array_equal(ih["toggle_list"], idv["toggle_list"]) and np.array_equal(ih["near_starts"], idv["near_starts"])
        n_done += int(dd.sum())
    assert n_done > 0
    # golden single-agent lap episode through the device logic
    e = gold("env_episode")
    env = amd.F110VecEnv(1, device_logic=True, obs_fields=("poses_x", "poses_y", "collisions"),
                         map=os.path.join(MAPS, "example_map"), map_ext='.png', num_agents=1, seed=12345)
    obs, r, done, info = env.reset(e["start"].reshape(1, 1, 3))
    for t, a in enumerate(e["actions"]):
        obs, r, done, info = env.step(a.reshape(1, 1, 2))
        k = t + 1
        assert float(info['toggle_list'][0, 0]) == e["toggle"][k] and bool(info['near_starts'][0, 0]) == bool(e["near"][k])
        assert float(obs['lap_counts'][0, 0]) == e["lap_count"][k] and bool(done[0]) == bool(e["done"][k])
        assert abs(float(obs['lap_times'][0, 0]) - e["lap_time"][k]) < 1e-12
    assert bool(done[0]) and "scans" not in obs


def test_waypoint_follow_two_laps_through_f110env(amd):
    """BASELINE configs[0]: the reference's waypoint_follow.py run (its planner's actions recorded
    from the reference) replayed through the drop-in F110Env for two laps, 3329 steps"""
    import os
    from _util import MAPS
    g = gold("waypoint_follow")
    env = amd.F110Env(map=os.path.join(MAPS, "example_map"), map_ext='.png', num_agents=1, timestep=0.01,
                      integrator=amd.Integrator.RK4)
    obs, r, done, info = env.reset(g["start"])
    worst = 0.0
    for t, a in enumerate(g["actions"]):
        obs, r, done, info = env.step(a.reshape(1, 2))
        row = g["traj"][t]
        worst = max(worst, rel_err([obs['poses_x'][0], obs['poses_y'][0], obs['poses_theta'][0], obs['linear_vels_x'][0],
                                    obs['ang_vels_z'][0]], row[:5]))
        assert float(obs['lap_counts'][0]) == row[6] and float(obs['collisions'][0]) == row[7] and done == bool(row[8]), t
        assert abs(float(obs['lap_times'][0]) - row[5]) < 1e-9
        if t % 32 == 0:
            assert abs(np.sum(obs['scans'][0]) - row[9]) / row[9] < NORTH_STAR
    assert worst < NORTH_STAR, worst
    assert done and float(obs['lap_counts'][0]) == 2.0


def test_step_other_maps_params_euler_lidar_offset(amd, orc):
    """step-level parity away from the bench defaults: berlin / skirk (resolution 0.05 is not a
    power of two -> guarded-division cell index; dt[-1,-1] = 0 -> rays end where they leave the
    map), Euler integrator, lidar offset, per-agent vehicle params, noise table shorter than the
    episode (wraps)"""
    rng = np.random.default_rng(17)
    for mapname, integ, ld, layout in [("berlin", 1, 0.0, 0), ("skirk", 2, 0.275, 0), ("berlin", 1, 0.275, 3), ("skirk", 1, 0.275, 3), ("berlin", 2, 0.0, 3)]:
        img, res, origin = load_map_image(mapname)
        dt, _, _ = oracle_map_dt(mapname)
        E, A, T = 9, 3, 50
        noise = _noise(7)                      # 7 rows only: step counts wrap modulo 7
        p2 = dict(amd.DEFAULT_PARAMS); p2.update({'mu': 0.8, 'm': 3.2, 'length': 0.50, 'width': 0.28, 'a_max': 7.0})
        s = amd.BatchSim(num_envs=E, num_agents=A, integrator=integ, lidar_dist=ld, map_layout=layout)
        s.set_map_image(img, res, origin); s.set_noise_table(noise); s.set_params(p2, 1)
        ref = orc.SimOracle(E, A, integrator=integ, lidar_dist=ld)
        ref.set_map_dt(dt, res, origin); ref.set_noise(noise); ref.set_params(p2, 1)
        with pytest.raises(IndexError):
            s.set_params(p2, 3)
        # cars start near the map's (0,0), which is free space on both maps, in a loose cluster
        poses = np.stack([rng.uniform(-0.6, 0.6, E * A), rng.uniform(-0.6, 0.6, E * A), rng.uniform(0, 2 * np.pi, E * A)], axis=1)
        s.reset(poses); ref.reset(poses)
        mism, es, er = 0, 0.0, 0.0
        for t in range(T):
            if t % 10 == 0:
                act = np.stack([rng.uniform(-0.4, 0.4, E * A), rng.uniform(-2.0, 5.0, E * A)], axis=1)
            s.step(act); ref.step(act, 8)
            o = s.get("scans", "state", "collisions", "collision_idx", "in_collision", "step_count")
            mism += int(np.sum(o["collisions"] != ref.collisions) + np.sum(o["in_collision"] != ref.in_collision)
                        + np.sum(o["collision_idx"] != ref.collision_idx))
            es = max(es, rel_err(o["state"], ref.state)); er = max(er, rel_err(o["scans"], ref.scans))
        assert mism == 0 and es < NORTH_STAR and er < NORTH_STAR, (mapname, mism, es, er)
        assert ref.collisions.sum() > 0          # the cluster does produce body collisions
        s.close()


@pytest.mark.gpu
def test_padded_layout_guard_band_and_far_poses(amd, orc):
    """PADDED layout on the device: lidars on cell corners with axis-aligned beams (guard band),
    on / just off / far off the map (border, then the exact fallback), all bit-equal to the oracle"""
    so = orc.ScanOracle(1080, 4.7)
    rng = np.random.default_rng(21)
    for mapname in ("berlin", "example_map", "skirk"):
        dt, res, origin = oracle_map_dt(mapname)
        so.set_map_dt(dt, res, origin)
        H, W = dt.shape
        free = np.argwhere(dt > 0.3)
        poses = []
        for r, c in free[rng.choice(len(free), 40, replace=False)]:
            poses.append([origin[0] + c * res, origin[1] + r * res, rng.choice([0.0, np.pi / 2, np.pi, -np.pi / 2])])
            poses.append([origin[0] + c * res, origin[1] + (r + 0.5) * res, 0.0])
            # beam 0 takes table direction 0 = (1, 0) exactly: it runs along the cell boundary y = const
            poses.append([origin[0] + (c + 0.25) * res, origin[1] + r * res, 4.7 / 2 + 1e-5])
            poses.append([origin[0] + (c + rng.uniform()) * res, origin[1] + (r + rng.uniform()) * res, rng.uniform(-7, 7)])
        poses += [[origin[0], origin[1], 0.3], [origin[0] - 1.0, origin[1] + H * res / 2, 0.0],
                  [origin[0] + W * res + 2.5, origin[1] + H * res + 2.5, 3.9], [origin[0] - 40.0, origin[1] - 40.0, 0.8],
                  [origin[0] + W * res / 2, origin[1] + H * res + 29.0, -1.6], [1e9, -1e9, 1.0], [1e300, 0.0, 0.0]]
        poses = np.asarray(poses)
        s = amd.BatchSim(num_envs=1, num_agents=1, map_layout=3)
        s.set_map_dt(dt, res, origin)
        s.scan_path_stats(enable=True)
        ranges, hits, lk = s.scan_batch(poses, want_hits=True, want_lookups=True)
        st = s.scan_path_stats()
        # the fixed-point march is what runs (no silent fallback); the rays that run along a cell
        # boundary are re-marched exactly; the 4 far-off lidars take the exact march
        assert st["fast"] + st["guard"] + st["exact"] == poses.shape[0] * 1080
        assert st["fast"] >= 160 * 1080 * 0.99 and 40 <= st["guard"] < 1e-3 * st["fast"] and st["exact"] >= 4 * 1080, st
        for k, pose in enumerate(poses):
            ref, ref_hits = so.scan(pose, want_hits=True)
            assert np.array_equal(hits[k], ref_hits), (mapname, pose)
            assert np.array_equal(ranges[k], ref), (mapname, pose)
            assert lk[k] == so.last_lookups
        s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("A", [1, 2, 3])
def test_auto_reseat_equals_step_plus_reset_collided(amd, A):
    """f110_set_auto_reseat (re-seat folded into the step's last kernel) leaves exactly the state
    that step + f110_reset_collided_device leaves, step after step, with envs finishing all the time"""
    E, T = 96, 120
    img, res, origin = load_map_image("example_map")
    w = raceline()
    rng = np.random.default_rng(5)
    k = rng.integers(0, w.shape[0], E)
    base = np.stack([w[k, 1], w[k, 2], w[k, 3] + np.pi / 2], axis=1)
    poses = np.repeat(base, A, axis=0) + np.stack([rng.uniform(-0.6, 0.6, E * A), rng.uniform(-0.6, 0.6, E * A), rng.uniform(-0.4, 0.4, E * A)], axis=1)
    sims, starts, counts = [], [], []
    for fused in (False, True):
        s = amd.BatchSim(num_envs=E, num_agents=A)
        s.set_map_image(img, res, origin)
        s.set_noise_table(np.random.default_rng(9).normal(0., 0.01, size=(T + 2, 1080)))
        d_start = s.device_array((E * A, 3)); d_start.upload(poses)
        d_count = s.device_array((1,), dtype=np.int32); d_count.upload(np.zeros(1, dtype=np.int32))
        s.reset_device(d_start)
        if fused:
            s.set_auto_reseat(d_start, A - 1, d_count)
        sims.append(s); starts.append(d_start); counts.append(d_count)
    d_act = [s.device_array((E * A, 2)) for s in sims]
    for t in range(T):
        act = np.stack([rng.uniform(-0.45, 0.45, E * A), rng.uniform(1.0, 9.0, E * A)], axis=1)
        for s, da, ds, dc, fused in zip(sims, d_act, starts, counts, (False, True)):
            da.upload(act)
            s.step_device(da)
            if not fused:
                s.reset_collided_device(ds, A - 1, dc)
        a, b = (s.get("scans", "state", "collisions", "collision_idx", "step_count") for s in sims)
        for key in a:
            assert np.array_equal(a[key], b[key]), (t, key)
    n = [int(c.download()[0]) for c in counts]
    assert n[0] == n[1] and n[0] > 10, n
    sims[1].set_auto_reseat(None)
    for s in sims:
        s.close()


@pytest.mark.gpu
def test_padded_layout_equals_rowmajor_at_scale(amd):
    """PADDED (fixed-point march + rare exact re-march) against the plain row-major kernel on half a
    billion rays: every scan value, flag and state bit-identical.  Large enough that some rays take
    the exact re-march (about one in 10^7) without being constructed for it."""
    E, A, T = 8192, 2, 30
    img, res, origin = load_map_image("example_map")
    poses = bench_start_poses(E, A)
    rng = np.random.default_rng(77)
    poses = poses + np.stack([rng.uniform(-0.3, 0.3, E * A), rng.uniform(-0.3, 0.3, E * A), rng.uniform(-0.3, 0.3, E * A)], axis=1)
    sims = []
    for layout in (3, 0):
        s = amd.BatchSim(num_envs=E, num_agents=A, map_layout=layout)
        s.set_map_image(img, res, origin)
        s.set_noise_table(np.random.default_rng(3).normal(0., 0.01, size=(T + 2, 1080)))
        s.reset(poses)
        sims.append(s)
    unit = amd.BatchSim(num_envs=1, num_agents=1, map_layout=3)
    unit.set_map_image(img, res, origin)
    unit.scan_path_stats(enable=True)
    redo = 0
    for t in range(T):
        act = np.stack([rng.uniform(-0.3, 0.3, E * A), rng.uniform(1.0, 7.0, E * A)], axis=1)
        for s in sims:
            s.step(act)
        if t % 3 == 2 or t == T - 1:
            a, b = (s.get("scans", "state", "collisions", "in_collision", "collision_idx") for s in sims)
            for key in a:
                assert np.array_equal(a[key], b[key]), (t, key)
            # the same lidar poses through the unit kernel, which reports the path each ray took
            st = a["state"]
            lid = np.stack([st[:, 0], st[:, 1], st[:, 4]], axis=1)
            r3 = unit.scan_batch(lid)
            redo += unit.scan_path_stats()["guard"]
    assert redo > 0, "no ray took the exact re-march: the test lost its point"
    for s in sims + [unit]:
        s.close()


@pytest.mark.gpu
def test_pure_pursuit_planner_vs_reference_and_oracle(amd, orc):
    """examples/waypoint_follow.py's PurePursuitPlanner through the C ABI: the reference's own
    outputs (golden, 775 poses) and the oracle on random poses"""
    g = gold("planner")
    wp = g["waypoints"]
    L, vg, wb = float(g["tlad"][0]), float(g["vgain"][0]), float(g["wheelbase"][0])
    pl = amd.PurePursuitPlanner(wp, wb)
    act = pl.plan_batch(g["poses"], L, vg)
    assert np.max(np.abs(act - g["actions"])) < 1e-13
    rng = np.random.default_rng(31)
    extra = np.stack([rng.uniform(-60, 20, 4000), rng.uniform(-30, 30, 4000), rng.uniform(-7, 7, 4000)], axis=1)
    act = pl.plan_batch(extra, L, vg)
    ref = np.array([orc.pure_pursuit_plan(wp, p, L, vg, wb) for p in extra])
    assert np.max(np.abs(act - ref)) < 1e-13
    # the reference's call signature: (speed, steer) for one pose
    sp, st = pl.plan(g["poses"][5, 0], g["poses"][5, 1], g["poses"][5, 2], L, vg)
    assert abs(st - g["actions"][5, 0]) < 1e-13 and abs(sp - g["actions"][5, 1]) < 1e-13
    pl.close()


@pytest.mark.gpu
def test_closed_loop_pure_pursuit_on_device(amd, orc):
    """plan -> step -> plan ... entirely on the device (actions never visit the host), 64 single-car
    envs spread around the raceline, against the oracle's planner + simulator in the same loop.
    Also the property the example shows: the planner laps the track without touching a wall."""
    g = gold("planner")
    wp = g["waypoints"]
    L, vg, wb = float(g["tlad"][0]), float(g["vgain"][0]), float(g["wheelbase"][0])
    E, T = 64, 400
    img, res, origin = load_map_image("example_map")
    dt, _, _ = oracle_map_dt("example_map")
    w = raceline()
    k = (np.arange(E) * 12) % w.shape[0]
    poses = np.stack([w[k, 1], w[k, 2], w[k, 3] + np.pi / 2], axis=1)
    s = amd.BatchSim(num_envs=E, num_agents=1)
    s.set_map_image(img, res, origin)
    ref = orc.SimOracle(E, 1)
    ref.set_map_dt(dt, res, origin)
    s.reset(poses)
    ref.reset(poses)
    pl = amd.PurePursuitPlanner(wp, wb, sim=s)
    d_act = s.device_array((E, 2))
    for t in range(T):
        pl.plan_device(s, d_act, L, vg)
        s.step_device(d_act)
        st_ref = ref.state
        act_ref = np.array([orc.pure_pursuit_plan(wp, [st_ref[e, 0], st_ref[e, 1], st_ref[e, 4]], L, vg, wb) for e in range(E)])
        ref.step(act_ref)
        if t % 50 == 49 or t == T - 1:
            o = s.get("state", "collisions", "scans")
            assert np.array_equal(o["collisions"], ref.collisions)
            assert rel_err(o["state"], ref.state) < 1e-6       # closed loop: differences feed back
            assert rel_err(o["scans"], ref.scans) < 1e-5
            assert np.max(np.abs(d_act.download() - act_ref)) < 1e-6
    # the example's tuning (vgain 1.375) is for its own start pose; from a standing start at an
    # arbitrary waypoint a few cars clip a wall — in the oracle too (asserted above) — the rest race
    racing = (o["collisions"] == 0) & (np.abs(o["state"][:, 3]) > 3.0)
    assert racing.sum() >= 0.9 * E
    pl.close()
    s.close()


@pytest.mark.gpu
def test_example_waypoint_follow_reproduces_the_reference_run():
    """examples/waypoint_follow.py (drop-in F110Env + the device planner, closed loop, nothing
    replayed) ends where the reference's own run of its example ended: same number of steps, two
    laps, same lap time, no collision (golden: tests/golden/waypoint_follow.npz)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("amd_waypoint_follow", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "waypoint_follow.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    res = ex.run(ex.load_conf())
    g = gold("waypoint_follow")
    assert res["done"] and not res["collided"]
    assert res["steps"] == g["actions"].shape[0]
    assert res["lap_count"] == g["traj"][-1, 6] == 2.0
    assert abs(res["lap_time"] - g["traj"][-1, 5]) < 1e-9


@pytest.mark.gpu
def test_per_env_maps_equal_single_map_sims(amd, orc):
    """f110_set_env_maps (a different track per env): every env group of a 3-map batch evolves bit
    for bit like a single-map simulator of that track, and like the oracle"""
    names = ["example_map", "berlin", "skirk"]
    maps = [load_map_image(n) for n in names]
    E, A, T = 30, 2, 60
    env_map = np.arange(E) % 3
    rng = np.random.default_rng(8)
    w = raceline()
    poses = np.zeros((E, A, 3))
    for e in range(E):
        if env_map[e] == 0:
            k = rng.integers(0, w.shape[0]); base = np.array([w[k, 1], w[k, 2], w[k, 3] + np.pi / 2])
        else:
            base = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(0, 6.28)])
        for a in range(A):
            poses[e, a] = base + np.array([rng.uniform(-0.7, 0.7), rng.uniform(-0.7, 0.7), rng.uniform(-0.4, 0.4)])
    poses = poses.reshape(E * A, 3)
    noise = np.random.default_rng(4).normal(0., 0.01, size=(T + 2, 1080))
    multi = amd.BatchSim(num_envs=E, num_agents=A)
    multi.set_map_image(*maps[0])
    assert multi.add_map_image(*maps[1]) == 1 and multi.add_map_image(*maps[2]) == 2
    multi.set_env_maps(env_map)
    multi.set_noise_table(noise)
    multi.reset(poses)
    singles, refs, sel = [], [], []
    for m in range(3):
        idx = np.where(env_map == m)[0]
        ag = (idx[:, None] * A + np.arange(A)[None, :]).reshape(-1)
        s = amd.BatchSim(num_envs=len(idx), num_agents=A)
        s.set_map_image(*maps[m]); s.set_noise_table(noise); s.reset(poses[ag])
        dt, res, origin = oracle_map_dt(names[m])
        r = orc.SimOracle(len(idx), A); r.set_map_dt(dt, res, origin); r.set_noise(noise); r.reset(poses[ag])
        singles.append(s); refs.append(r); sel.append(ag)
    for t in range(T):
        act = np.stack([rng.uniform(-0.4, 0.4, E * A), rng.uniform(0.5, 8.0, E * A)], axis=1)
        multi.step(act)
        for s, r, ag in zip(singles, refs, sel):
            s.step(act[ag]); r.step(act[ag])
        if t % 6 == 5 or t == T - 1:
            o = multi.get("scans", "state", "collisions", "in_collision", "collision_idx")
            for s, r, ag in zip(singles, refs, sel):
                q = s.get("scans", "state", "collisions", "in_collision", "collision_idx")
                for key in q:
                    assert np.array_equal(o[key][ag], q[key]), (t, key)
                assert np.array_equal(o["collisions"][ag], r.collisions) and np.array_equal(o["in_collision"][ag], r.in_collision)
                assert rel_err(o["state"][ag], r.state) < 1e-9 and rel_err(o["scans"][ag], r.scans) < 1e-9
    # back to one map for everybody
    multi.set_env_maps(None)
    with pytest.raises(Exception):
        multi.set_env_maps(np.full(E, 7))
    for s in singles + [multi]:
        s.close()


@pytest.mark.gpu
def test_vec_env_tracks_per_env(amd):
    """F110VecEnv(extra_maps=..., env_map=...): envs on berlin / skirk see those tracks' scans"""
    from _util import MAPS
    import os
    E = 6
    env = amd.F110VecEnv(E, map=os.path.join(MAPS, "berlin"), map_ext=".png", num_agents=1, scan_noise_std=0.0,
                         extra_maps=[(os.path.join(MAPS, "skirk.yaml"), ".png")], env_map=[0, 1, 0, 1, 0, 1])
    poses = np.tile(np.array([[[0.0, 0.0, 1.0]]]), (E, 1, 1))
    obs, _, _, _ = env.reset(poses)
    singles = {}
    for name in ("berlin", "skirk"):
        s = amd.ScanSimulator2D(1080, 4.7)
        s.set_map(os.path.join(MAPS, name + ".yaml"), ".png")
        singles[name] = s
    # reset() advances one zero-action step from rest: the pose is unchanged
    for e in range(E):
        ref = singles["berlin" if e % 2 == 0 else "skirk"].scan(np.array([0.0, 0.0, 1.0]), None)
        assert np.array_equal(obs["scans"][e, 0], ref), e
    assert not np.array_equal(obs["scans"][0, 0], obs["scans"][1, 0])


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [3, 0])
def test_step_on_rotated_non_pow2_map_vs_oracle(amd, orc, layout):
    """the step's scan kernels on a map whose origin is rotated and whose resolution is not a power
    of two (the generic position transform of the PADDED layout, the guarded division of the
    row-major one): cars driving on a cut-out of example_map re-expressed in such a frame"""
    dt, res, _ = oracle_map_dt("example_map")
    sub = np.ascontiguousarray(dt[500:1100, 800:1400])
    res2, org = 0.05, [-3.0, -4.0, 0.35]
    table = sub * (res2 / res)
    c, s_ = np.cos(org[2]), np.sin(org[2])
    E, A, T = 12, 2, 40
    rng = np.random.default_rng(17)
    free = np.argwhere(table > 0.6)
    pick = free[rng.choice(len(free), E, replace=False)]
    poses = np.zeros((E, A, 3))
    for e, (r, cc) in enumerate(pick):
        u, v = (cc + 0.5) * res2, (r + 0.5) * res2
        base = np.array([org[0] + c * u - s_ * v, org[1] + s_ * u + c * v, rng.uniform(0, 6.28)])
        for a in range(A):
            poses[e, a] = base + np.array([rng.uniform(-0.25, 0.25), rng.uniform(-0.25, 0.25), rng.uniform(-0.3, 0.3)])
    poses = poses.reshape(E * A, 3)
    noise = np.random.default_rng(2).normal(0., 0.01, size=(T + 2, 1080))
    sim = amd.BatchSim(num_envs=E, num_agents=A, map_layout=layout)
    sim.set_map_dt(table, res2, org)
    sim.set_noise_table(noise)
    ref = orc.SimOracle(E, A)
    ref.set_map_dt(table, res2, org)
    ref.set_noise(noise)
    sim.reset(poses); ref.reset(poses)
    for t in range(T):
        act = np.stack([rng.uniform(-0.4, 0.4, E * A), rng.uniform(0.5, 6.0, E * A)], axis=1)
        sim.step(act); ref.step(act)
        if t % 5 == 4 or t == T - 1:
            o = sim.get("scans", "state", "collisions", "in_collision", "collision_idx")
            assert np.array_equal(o["collisions"], ref.collisions) and np.array_equal(o["in_collision"], ref.in_collision)
            assert np.array_equal(o["collision_idx"], ref.collision_idx)
            assert rel_err(o["state"], ref.state) < 1e-9 and rel_err(o["scans"], ref.scans) < 1e-9
    sim.close()


@pytest.mark.gpu
def test_per_agent_vehicle_params(amd, orc):
    """f110_set_params_batch (a parameter set per agent, for domain randomisation): every env evolves
    like a single-env simulator holding that env's sets in its agent slots, and like the oracle"""
    E, A, T = 5, 2, 80
    img, res, origin = load_map_image("example_map")
    dt, _, _ = oracle_map_dt("example_map")
    rng = np.random.default_rng(12)
    sets = []
    for i in range(E * A):
        p = dict(amd.DEFAULT_PARAMS)
        p.update({'mu': rng.uniform(0.6, 1.2), 'm': rng.uniform(3.0, 4.2), 'lf': rng.uniform(0.147, 0.17), 'C_Sf': rng.uniform(4.0, 5.5),
                  'a_max': rng.uniform(7.0, 10.0), 'v_max': rng.uniform(12.0, 22.0), 'length': rng.uniform(0.5, 0.62), 'width': rng.uniform(0.27, 0.34)})
        sets.append(p)
    poses = bench_start_poses(E, A) + np.stack([rng.uniform(-0.3, 0.3, E * A), rng.uniform(-0.3, 0.3, E * A), rng.uniform(-0.3, 0.3, E * A)], axis=1)
    noise = np.random.default_rng(6).normal(0., 0.01, size=(T + 2, 1080))
    batch = amd.BatchSim(num_envs=E, num_agents=A)
    batch.set_map_image(img, res, origin); batch.set_noise_table(noise)
    batch.set_params_batch(sets)
    with pytest.raises(Exception):
        batch.set_params(sets[0], 0)
    batch.reset(poses)
    singles, refs = [], []
    for e in range(E):
        s = amd.BatchSim(num_envs=1, num_agents=A); s.set_map_image(img, res, origin); s.set_noise_table(noise)
        r = orc.SimOracle(1, A); r.set_map_dt(dt, res, origin); r.set_noise(noise)
        for a in range(A):
            s.set_params(sets[e * A + a], a); r.set_params(sets[e * A + a], a)
        s.reset(poses[e * A:(e + 1) * A]); r.reset(poses[e * A:(e + 1) * A])
        singles.append(s); refs.append(r)
    for t in range(T):
        act = np.stack([rng.uniform(-0.4, 0.4, E * A), rng.uniform(0.5, 9.0, E * A)], axis=1)
        batch.step(act)
        for e in range(E):
            singles[e].step(act[e * A:(e + 1) * A]); refs[e].step(act[e * A:(e + 1) * A])
        if t % 8 == 7 or t == T - 1:
            o = batch.get("scans", "state", "collisions", "in_collision", "collision_idx")
            for e in range(E):
                q = singles[e].get("scans", "state", "collisions", "in_collision", "collision_idx")
                for key in q:
                    assert np.array_equal(o[key][e * A:(e + 1) * A], q[key]), (t, e, key)
                assert np.array_equal(o["collisions"][e * A:(e + 1) * A], refs[e].collisions)
                assert rel_err(o["state"][e * A:(e + 1) * A], refs[e].state) < 1e-9
                assert rel_err(o["scans"][e * A:(e + 1) * A], refs[e].scans) < 1e-9
    batch.set_params_batch(None)
    for s in singles + [batch]:
        s.close()
