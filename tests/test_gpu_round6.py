"""GPU tests added in round 6 (-m gpu): BASELINE configs[3] at its real size on one device (8 ranks as threads, the RCCL stand-in),
`bench.py --gpus 8 --ranks-in-process` through every gather leg at that size, `ShardedVecEnv` against a single handle, the fuzzers
at a few hundred seeds each, non-uniform `scan_angles` in ray_cast / check_ttc_jit against a reference-generated
fixture, and the rest of the functions `from f110_gym.envs import *` exposes in the reference against reference-generated rows.
Nothing here reads /root/reference."""
import importlib.util
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from _util import bench_start_poses, map_stem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_DIR = os.path.join(ROOT, "tests", "rccl_stub")


@pytest.fixture(scope="module")
def amd():
    import f1tenth_gym_amd
    from f1tenth_gym_amd import _ffi
    assert _ffi.device_count() >= 1, "no MI355X visible: the HIP path cannot run (no CPU fallback)"
    return f1tenth_gym_amd


def _stub_env():
    """environment in which `librccl.so.1` resolves to tests/rccl_stub (built here if stale)"""
    lib, src = os.path.join(STUB_DIR, "librccl.so.1"), os.path.join(STUB_DIR, "rccl_stub.hip")
    if not os.path.isfile(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", lib])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "F110_BENCH_RDV")}
    env["LD_LIBRARY_PATH"] = STUB_DIR + os.pathsep + os.environ.get("LD_LIBRARY_PATH", "")
    return env


def _result(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
    assert lines, (out.stdout[-1500:], out.stderr[-1500:])
    return json.loads(lines[0][7:])


# ------------------------------------------------------------------ BASELINE configs[3] at its size, one device
def test_config3_full_size_eight_ranks_on_one_device(amd):
    """262 144 agents = 8 ranks x 16 384 envs x 2, 1080 beams: every rank checks every peer's 283 MB block of its 2.26 GB receive
    buffers (digests), the twin envs across block boundaries, and its first 32 envs against the oracle — in the step's stream,
    overlapped (double-buffered) and as float32 to one root (tests/rccl_stub/config3_full_size.py)"""
    out = subprocess.run([sys.executable, os.path.join(STUB_DIR, "config3_full_size.py"), "8", "16384"], env=_stub_env(), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=900)
    r = _result(out)
    assert not r["errors"] and not any(r["hung"]) and out.returncode == 0, (r, out.stderr[-1500:])
    assert r["agents_total"] == 262144 and r["world"] == 8
    assert min(r["checks"]) > 0 and r["checks"][0] >= 3 * r["steps"] * 8     # rank 0 receives in all three legs


def test_bench_config3_eight_ranks_in_process_all_gather_legs(amd):
    """`python bench.py --gpus 8 --ranks-in-process --agents 32768` on device 0 with the RCCL stand-in: the bench's own Workload,
    control plane, leg records and digests at BASELINE configs[3]'s size — the headline without a collective and all six gather legs
    (in stream, overlapped, float32, float32 overlapped, to a root, root + float32 + overlapped), `gather_ok` on every rank"""
    env = _stub_env()
    env["F110_BENCH_DEVICE"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--ranks-in-process", "--agents", "32768", "--steps", "8", "--warmup", "2",
                          "--preroll", "40", "--gather-timeout", "300", "--gather-budget", "600"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-800:]
    d = json.loads(lines[0])
    mg = d["multi_gpu"]
    assert d["n_gpus"] == 8 and d["config"]["agents_total"] == 262144 and d["scaling"] == "weak"
    assert mg.get("gather_error") is None and not mg.get("legs_skipped"), mg
    assert mg["rccl_ranks"] == 8 and "threads" in mg["ranks_are"]
    for leg in ("gather", "gather_overlap", "gather_f32", "gather_f32_overlap", "gather_root", "gather_root_f32_overlap"):
        assert mg[leg]["gather_ok"] is True and mg[leg]["value"] > 0 and mg[leg]["rccl_ranks"] == 8, (leg, mg[leg])
        assert len(mg[leg]["per_rank_ms_per_step"]) == 8
    assert mg["gather"]["bytes_received_per_step"]["every_rank"] == 32768 * (8 * 1080 + 56) * 8
    assert mg["gather"]["device_mem_used_gb_max"] > 8 * 2.26     # the eight [8][32768][1080] float64 receive buffers were really there


# ------------------------------------------------------------------ ShardedVecEnv
@pytest.mark.parametrize("device_logic", [True, False])
@pytest.mark.parametrize("devices,sizes", [([0], None), ([0, 0, 0], None), ([0, 0, 0, 0, 0], [1, 9, 3, 20, 4])])
def test_sharded_vec_env_equals_one_handle(amd, devices, sizes, device_logic):
    """ShardedVecEnv over several handles (here all on device 0; equal, uneven and hand-picked shard sizes) returns bit for bit
    what ONE F110VecEnv returns for the same envs: observations, done, lap bookkeeping, through auto-resets and a partial reset"""
    E, A, T = 37, 2, 70
    kw = dict(map=map_stem("example_map"), map_ext=".png", num_agents=A, auto_reset=True, device_logic=device_logic)
    one = amd.F110VecEnv(E, **kw)
    # (both ways of reaching the shards: enqueue-all-then-wait-all from the caller's thread, and one worker thread per shard)
    sh = amd.ShardedVecEnv(E, devices=devices, shard_sizes=sizes, threaded_step=(len(devices) == 3), **kw)
    assert sh.threaded_step == (len(devices) == 3 or not device_logic)
    assert sh.shard_sizes == (sizes or [E // len(devices) + (1 if k < E % len(devices) else 0) for k in range(len(devices))])
    poses = bench_start_poses(E, A, gap_wp=4).reshape(E, A, 3)

    def same(a, b, what):
        for key in a[0]:
            assert np.array_equal(np.asarray(a[0][key]), np.asarray(b[0][key])), (what, key)
        assert a[1] == b[1] and np.array_equal(a[2], b[2]), (what, "done")
        assert set(a[3]) == set(b[3])
        for key in a[3]:
            assert np.array_equal(a[3][key], b[3][key]), (what, key)
    same(one.reset(poses), sh.reset(poses), "reset")
    rng = np.random.default_rng(5)
    seen_done = 0
    for t in range(T):
        act = np.stack([rng.uniform(-0.4, 0.4, (E, A)), rng.uniform(2.0, 8.0, (E, A))], axis=2)
        a, b = one.step(act), sh.step(act)
        same(a, b, "step %d" % t)
        seen_done += int(np.sum(a[2]))
        if t == 30:
            mask = np.arange(E) % 3 == 1
            same(one.reset(poses, mask), sh.reset(poses, mask), "partial reset")
    assert seen_done > 0      # the comparison went through re-seats
    sh.close(); one.sim.batch.close()


def test_sharded_vec_env_big_observation_blocks(amd):
    """600 envs over three handles: 10 MB of scans per step, which the shards' worker threads copy into the assembled arrays
    themselves (the parallel branch of ShardedVecEnv._assemble) — against one handle"""
    E, A = 600, 2
    kw = dict(map=map_stem("example_map"), map_ext=".png", num_agents=A, auto_reset=True)
    one = amd.F110VecEnv(E, device_logic=True, **kw)
    sh = amd.ShardedVecEnv(E, devices=[0, 0, 0], **kw)
    poses = bench_start_poses(E, A, gap_wp=4).reshape(E, A, 3)
    a, b = one.reset(poses), sh.reset(poses)
    rng = np.random.default_rng(6)
    for t in range(8):
        for key in a[0]:
            assert np.array_equal(np.asarray(a[0][key]), np.asarray(b[0][key])), (t, key)
        assert np.array_equal(a[2], b[2])
        act = np.stack([rng.uniform(-0.4, 0.4, (E, A)), rng.uniform(2.0, 8.0, (E, A))], axis=2)
        a, b = one.step(act), sh.step(act)
    sh.close(); one.sim.batch.close()


def test_example_sharded_vec_env_runs(amd):
    """examples/sharded_vec_env.py: two handles on device 0, with and without bringing the scans to the host"""
    ex = os.path.join(ROOT, "examples", "sharded_vec_env.py")
    for extra in ([], ["--scans"]):
        out = subprocess.run([sys.executable, ex, "--envs", "256", "--devices", "0,0", "--steps", "40"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             text=True, timeout=600)
        assert out.returncode == 0 and "agent-steps/s" in out.stdout and "2 handle(s)" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])


def test_sharded_vec_env_gathers_the_observation_on_one_device(amd):
    """gather_obs=True: after each step every shard's device holds every shard's scans + scalars = the single handle's observation in
    blocks (all-gather float64, and float32 to one root) — four handles on device 0, the RCCL stand-in (tests/rccl_stub/sharded_gather.py)"""
    out = subprocess.run([sys.executable, os.path.join(STUB_DIR, "sharded_gather.py")], env=_stub_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=600)
    r = _result(out)
    assert not r["errors"] and out.returncode == 0, (r, out.stderr[-1500:])
    assert r["checks"] == 12 * 4 + 12 * 1      # 12 steps x 4 receiving shards (all-gather), then 12 x the one root


def test_sharded_vec_env_gather_through_real_rccl_world_size_one(amd):
    """the same gather path on the REAL librccl (a communicator of one rank is all one device allows): ShardedVecEnv(devices=[0],
    gather_obs=True) — comm init from the worker thread, the all-gather on the shard's stream, gathered_views() == the step's observation"""
    E, A = 12, 2
    sh = amd.ShardedVecEnv(E, devices=[0], gather_obs=True, map=map_stem("example_map"), map_ext=".png", num_agents=A,
                           obs_fields=("scans", "poses_x", "poses_y", "poses_theta", "linear_vels_x", "ang_vels_z", "collisions"))
    assert sh.shards[0].sim.batch.comm_info() == (1, 0)
    obs = sh.reset(bench_start_poses(E, A).reshape(E, A, 3))[0]
    rng = np.random.default_rng(4)
    for t in range(8):
        obs = sh.step(np.stack([rng.uniform(-0.3, 0.3, (E, A)), rng.uniform(1.0, 6.0, (E, A))], axis=2))[0]
        sh.sync()
        d_s, d_c = sh.gathered_views()[0]
        assert np.array_equal(d_s.download()[0], obs["scans"].reshape(E * A, -1)), t
        want = np.stack([obs[k].reshape(-1) for k in ("poses_x", "poses_y", "poses_theta", "linear_vels_x")] + [np.zeros(E * A)]
                        + [obs[k].reshape(-1) for k in ("ang_vels_z", "collisions")])
        assert np.array_equal(d_c.download()[0], want), t
    sh.close()


# ------------------------------------------------------------------ the rest of `from f110_gym.envs import *` (star_exports.npz: reference-run)
FTOL = 1e-12


def _rel(a, b):
    from _util import rel_err
    return rel_err(a, b)


def test_star_exported_helpers_vs_reference(amd):
    """accl_constraints / steering_constraint (dynamic_models.py:29-87), cross / are_collinear (laser_models.py:219-247), perpendicular /
    tripleProduct / avgPoint / indexOfFurthestPoint / support / get_trmtx (collision_models.py:34-110, :218-235): f110_helper_batch
    against rows the reference produced — exact (sums and products in the reference's order), except tripleProduct (its dot products are
    BLAS calls in the reference: 1e-9) and get_trmtx (cos / sin ulps: 1e-12)"""
    from _util import gold
    from f1tenth_gym_amd import _ffi
    g = gold("star_exports")
    b = amd.BatchSim(num_envs=1, num_agents=1)
    assert np.array_equal(b.helper_batch(_ffi.OP_ACCL_CONSTRAINTS, g["accl_in"])[:, 0], g["accl_out"])
    assert np.array_equal(b.helper_batch(_ffi.OP_STEERING_CONSTRAINT, g["steer_in"])[:, 0], g["steer_out"])
    assert np.array_equal(b.helper_batch(_ffi.OP_CROSS, g["cross_in"])[:, 0], g["cross_out"])
    assert np.array_equal(b.helper_batch(_ffi.OP_ARE_COLLINEAR, g["collinear_in"])[:, 0], g["collinear_out"])
    assert 0 < g["collinear_out"].sum() < len(g["collinear_out"])
    assert np.array_equal(b.helper_batch(_ffi.OP_PERPENDICULAR, g["perp_in"]), g["perp_out"])
    # tripleProduct: the reference's two a.dot(c) go through BLAS (fused multiply-adds there, plain multiply + add here): ulps
    assert np.allclose(b.helper_batch(_ffi.OP_TRIPLE_PRODUCT, g["triple_in"]), g["triple_out"], rtol=1e-9, atol=1e-12)
    n = len(g["body_a"])
    va, vb, d = g["body_a"].reshape(n, 8), g["body_b"].reshape(n, 8), g["dir"]
    assert np.array_equal(b.helper_batch(_ffi.OP_AVG_POINT, va, n=4), g["avg_out"])
    # indexOfFurthestPoint / support: argmax of vertices.dot(d).  Where two vertices project equally far (the fixture has such rows
    # on purpose: d = 0, d along a box edge) the winner is decided by the last bit of a BLAS dot product in the reference; there the
    # device's choice must be A maximiser (to 1e-12), everywhere else THE reference's index
    def check_furthest(bodies, dirs, got, want, n):
        proj = np.einsum("mij,mj->mi", bodies.reshape(len(bodies), n, 2), dirs)
        top2 = np.sort(proj, axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 1e-9 * (1.0 + np.abs(top2[:, 1]))
        assert clear.sum() > 0.6 * len(bodies) and (~clear).sum() > 0
        assert np.array_equal(got[clear], want[clear])
        assert np.all(proj[np.arange(len(bodies)), got] >= proj.max(axis=1) - 1e-12 * (1.0 + np.abs(proj.max(axis=1))))
        return clear
    got = b.helper_batch(_ffi.OP_FURTHEST_POINT, np.concatenate([va, d], axis=1), n=4)[:, 0].astype(np.int32)
    clear_a = check_furthest(va, d, got, g["furthest_out"], 4)
    clear_b = check_furthest(vb, -d, b.helper_batch(_ffi.OP_FURTHEST_POINT, np.concatenate([vb, -d], axis=1), n=4)[:, 0].astype(np.int32),
                             np.array([int(np.argmax(vb[i].reshape(4, 2).dot(-d[i]))) for i in range(n)]), 4)
    sup = b.helper_batch(_ffi.OP_SUPPORT, np.concatenate([va, vb, d], axis=1), n=4)
    both = clear_a & clear_b
    assert np.array_equal(sup[both], g["support_out"][both])
    assert np.all(np.abs(np.einsum("mj,mj->m", sup - g["support_out"], d)) <= 1e-12 * (1.0 + np.abs(np.einsum("mj,mj->m", sup, d))))   # equally far along d everywhere
    pent = g["pent"].reshape(-1, 10)
    assert np.array_equal(b.helper_batch(_ffi.OP_AVG_POINT, pent, n=5), g["pent_avg"])
    assert np.array_equal(b.helper_batch(_ffi.OP_FURTHEST_POINT, np.concatenate([pent, g["pent_dir"]], axis=1), n=5)[:, 0].astype(np.int32), g["pent_furthest"])
    assert _rel(b.helper_batch(_ffi.OP_GET_TRMTX, g["trmtx_in"]).reshape(-1, 4, 4), g["trmtx_out"]) < FTOL
    b.close()
    # and through the reference's own signatures (f110_gym.envs star-exports)
    import f110_gym.envs as envs
    assert envs.accl_constraints(*g["accl_in"][3]) == g["accl_out"][3] and envs.steering_constraint(*g["steer_in"][25]) == g["steer_out"][25]
    assert envs.cross(g["cross_in"][0, :2], g["cross_in"][0, 2:]) == g["cross_out"][0]
    r = g["collinear_in"][5]
    assert envs.are_collinear(r[0:2], r[2:4], r[4:6]) == bool(g["collinear_out"][5])
    pt = g["perp_in"][7].copy()
    assert envs.perpendicular(pt) is pt and np.array_equal(pt, g["perp_out"][7])
    r = g["triple_in"][9]
    assert np.allclose(envs.tripleProduct(r[0:2], r[2:4], r[4:6]), g["triple_out"][9], rtol=1e-9, atol=1e-12)
    assert np.array_equal(envs.avgPoint(g["body_a"][11]), g["avg_out"][11]) and np.array_equal(envs.avgPoint(g["pent"][3]), g["pent_avg"][3])
    assert envs.indexOfFurthestPoint(g["body_a"][200], g["dir"][200]) == g["furthest_out"][200]      # (rows from 120 on are random directions: no ties)
    assert np.array_equal(envs.support(g["body_a"][213], g["body_b"][213], g["dir"][213]), g["support_out"][213])
    assert _rel(envs.get_trmtx(g["trmtx_in"][2]), g["trmtx_out"][2]) < FTOL


@pytest.mark.parametrize("tag", ["a", "b"])
def test_free_scan_functions_vs_reference(amd, tag):
    """get_dt, xy_2_rc, distance_transform, trace_ray, get_scan as FREE functions with the reference's own argument lists
    (laser_models.py:40-186), on a synthetic map whose origin is plain (a) and yawed by 0.6 rad (b): cells exact, table values
    and ranges exact (the same additions in the same order; only the table's sqrt and the caller's sin / cos enter, both host-side)"""
    from _util import gold
    import f110_gym.envs as envs
    from f1tenth_gym_amd import _ffi, functional
    g = gold("star_exports")
    res = float(g["map_res"][0])
    dt = envs.get_dt(g["bitmap"].astype(np.float64), res)
    assert np.array_equal(dt, g["dt"])
    H, W = dt.shape
    ox, oy, oth = g["origin_" + tag]
    oc, os_ = np.cos(oth), np.sin(oth)
    pts = g["pts_" + tag]
    m = len(pts)
    b = amd.BatchSim(num_envs=1, num_agents=1)
    rows = np.concatenate([pts, np.tile([ox, oy, oc, os_, H, W, res], (m, 1))], axis=1)
    assert np.array_equal(b.helper_batch(_ffi.OP_XY_2_RC, rows).astype(np.int32), g["rc_" + tag])
    b.close()
    assert (g["rc_" + tag][:, 0] < 0).any() and (g["rc_" + tag][:, 0] >= 0).sum() > 100
    th = np.linspace(0.0, 2 * np.pi, num=720)
    sines, cosines = np.sin(th), np.cos(th)
    for i in list(range(0, m, 7)) + [0, 1, 2]:
        assert envs.xy_2_rc(pts[i, 0], pts[i, 1], ox, oy, oc, os_, H, W, res) == tuple(g["rc_" + tag][i])
        assert envs.distance_transform(pts[i, 0], pts[i, 1], ox, oy, oc, os_, H, W, res, dt) == g["dtval_" + tag][i]
        assert envs.trace_ray(pts[i, 0], pts[i, 1], g["theta_idx_" + tag][i], sines, cosines, 1e-4, ox, oy, oc, os_, H, W, res, dt, 4.0) == g["trace_" + tag][i]
    # every row in one launch each
    hb = functional._map_sim(dt, res, ox, oy, oc, os_, eps=1e-4, theta_dis=720, max_range=4.0, sines=sines, cosines=cosines)
    assert np.array_equal(hb.helper_batch(_ffi.OP_DISTANCE_TRANSFORM, pts)[:, 0], g["dtval_" + tag])
    assert np.array_equal(hb.helper_batch(_ffi.OP_TRACE_RAY, np.concatenate([pts, g["theta_idx_" + tag][:, None]], axis=1))[:, 0], g["trace_" + tag])
    theta_dis, nb, fov, eps, mr = g["scan_cfg"]
    inc = int(theta_dis) * (fov / (int(nb) - 1)) / (2. * np.pi)
    for q, want in zip(g["scan_poses_" + tag], g["scans_" + tag]):
        got = envs.get_scan(q, int(theta_dis), fov, int(nb), inc, sines, cosines, eps, ox, oy, oc, os_, H, W, res, dt, mr)
        assert np.array_equal(got, want)
    with pytest.raises(ValueError):
        envs.get_scan(g["scan_poses_" + tag][0], int(theta_dis), fov, int(nb), inc * 1.01, sines, cosines, eps, ox, oy, oc, os_, H, W, res, dt, mr)
    functional.close_cached_handles()


@pytest.mark.parametrize("table", ["two_ramps", "jitter", "descending", "short_random", "uniform_ref"])
def test_scan_angles_tables_that_are_not_the_uniform_ramp(amd, table):
    """ray_cast / get_blocked_view_indices / check_ttc_jit with ANY scan_angles array (laser_models.py:282-346, :188-217): two
    concatenated ramps, a jittered ramp, a descending ramp, 37 random angles — the reference's full first-minimum argmin decides the
    window there (k_raycast_unit's non-uniform branch); window exact, scans <= 1e-12, flags exact.  The step refuses such tables."""
    from _util import gold
    import f110_gym.envs as envs
    from f1tenth_gym_amd import functional
    g = gold("star_exports")
    sa = g["sa_" + table]
    B = len(sa)
    ego, verts = g["ego_" + table], g["verts_" + table]
    m = len(ego)
    b = amd.BatchSim(num_envs=1, num_agents=1, num_beams=B)
    b.set_beam_tables(sa, g["ttc_cos_" + table], g["ttc_side_" + table])
    out, mm = b.raycast_batch(ego, verts, np.full((m, B), 10.0))
    assert np.array_equal(mm[:, 0], g["lo_" + table]) and np.array_equal(mm[:, 1], g["hi_" + table])
    assert _rel(out, g["rc_scans_" + table]) < FTOL and (out < 10.0).sum() == (g["rc_scans_" + table] < 10.0).sum()
    assert np.array_equal(b.ttc_batch(g["ttc_scans_" + table], g["ttc_vel_" + table], 0.005), g["ttc_flags_" + table])
    if table != "uniform_ref":
        img, res, origin = __import__("_util").load_map_image("example_map")
        b.set_map_image(img, res, origin)
        b.reset(np.array([[0.7, 0.0, 1.37]]))
        with pytest.raises(Exception, match="uniform ramp"):
            b.step(np.zeros((1, 2)))
    b.close()
    # the reference's signatures, alternating the three functions on one table (cached handle, tables uploaded once)
    for i in (0, 7, 19, m - 1):
        assert envs.get_blocked_view_indices(ego[i], verts[i], sa) == (g["lo_" + table][i], g["hi_" + table][i])
        sc = np.full(B, 10.0)
        assert envs.ray_cast(ego[i], sc, sa, verts[i]) is sc and _rel(sc, g["rc_scans_" + table][i]) < FTOL
        j = i % len(g["ttc_vel_" + table])
        assert envs.check_ttc_jit(g["ttc_scans_" + table][j], g["ttc_vel_" + table][j], sa, g["ttc_cos_" + table], g["ttc_side_" + table], 0.005) == bool(g["ttc_flags_" + table][j])
    assert len(functional._ctx) >= 1
    functional.close_cached_handles()


# ------------------------------------------------------------------ the whole step of a tiny batch as ONE launch (k_step_tiny)
def _tiny_env(amd, tiny, E, A, **kw):
    os.environ["F110_EXP"] = "step_tiny=%d" % tiny        # (read by the host when the handle is made; the product library refuses it: lab-only test)
    try:
        return amd.F110VecEnv(E, map=map_stem(kw.pop("track", "example_map")), map_ext=".png", num_agents=A, **kw)
    finally:
        del os.environ["F110_EXP"]


@pytest.mark.parametrize("E,A,track,device_logic", [(1, 2, "example_map", True), (1, 1, "example_map", True), (2, 2, "example_map", True), (4, 1, "berlin", True),
                                                    (3, 1, "skirk", False), (1, 2, "berlin", False), (2, 2, "example_map", False)])
def test_tiny_step_equals_the_three_kernels(amd, E, A, track, device_logic):
    """k_step_tiny (a host-synchronised step of at most 4 agents, one or two per env: integrate + scan + finalize + observation block +
    completion word in ONE launch — the reference's own shape) against the three-kernel form of the same library (lab switch
    step_tiny = 0): every observation, `done`, the lap bookkeeping and the device-side state bit for bit over 400 steps with wall hits,
    contacts, noise from the device generator, auto-resets; the launch count says which form ran.  device_logic=False is Simulator.step's
    form (every column incl. the scans in the block, episode logic on the host), True is F110VecEnv's (episode logic in the kernel)"""
    T = 400
    recs, finals = [], []
    for tiny in (1, 0):
        env = _tiny_env(amd, tiny, E, A, track=track, auto_reset=True, device_logic=device_logic)
        rng = np.random.default_rng(9)
        if track == "example_map":
            poses = bench_start_poses(E, A, gap_wp=3).reshape(E, A, 3)
        else:
            poses = np.stack([rng.uniform(-0.6, 0.6, (E, A)), rng.uniform(-0.6, 0.6, (E, A)), rng.uniform(0, 6.28, (E, A))], axis=2)
        def snap(o):     # (device_logic=True hands out VIEWS of the page-locked block, overwritten by the next step: copy now)
            return ({k: np.array(v) for k, v in o[0].items()}, np.array(o[2]), {k: np.array(v) for k, v in o[3].items()})
        out = [snap(env.reset(poses))]
        for t in range(T):
            out.append(snap(env.step(np.stack([rng.uniform(-0.4, 0.4, (E, A)), rng.uniform(4.0, 12.0, (E, A))], axis=2))))     # fast and blind: wall hits, contacts, re-seats
            assert env.sim.batch.step_launches() == tiny, t
        recs.append(out)
        finals.append(env.sim.batch.get("scans", "state", "collisions", "collision_idx", "in_collision", "agent_poses", "step_count"))
        env.sim.batch.close()
    assert sum(int(r[1].sum()) for r in recs[0]) > 0       # episodes ended (and were re-seated) on the way
    for t, (ra, rb) in enumerate(zip(*recs)):
        for key in ra[0]:
            assert np.array_equal(ra[0][key], rb[0][key]), (t, key)
        assert np.array_equal(ra[1], rb[1]), (t, "done")
        for key in ra[2]:
            assert np.array_equal(ra[2][key], rb[2][key]), (t, key)
    for key in finals[0]:
        assert np.array_equal(finals[0][key], finals[1][key]), key


def test_which_steps_take_the_one_launch_form(amd):
    """the product's dispatch: a WAITING f110_step_host of at most 4 agents with one or two cars per env is ONE launch (F110Env.step,
    F110VecEnv.step on one or two envs); more agents, more cars per env, step_async, the device-resident entry points, profiling or a
    per-agent noise stream take the per-kernel form"""
    from _util import load_map_image
    kw = dict(map=map_stem("example_map"), map_ext=".png")
    for A in (1, 2):
        env = amd.F110Env(num_agents=A, **kw)
        env.reset(bench_start_poses(1, A).reshape(A, 3))
        env.step(np.tile([0.0, 2.0], (A, 1)))
        assert env.sim.batch.step_launches() == 1, A
        env.sim.batch.close()
    for E, A, want in ((2, 2, 1), (4, 1, 1), (3, 2, 0), (5, 1, 0), (1, 3, 0)):
        v = amd.F110VecEnv(E, num_agents=A, device_logic=True, **kw)
        v.reset(bench_start_poses(E, A).reshape(E, A, 3))
        v.step(np.tile([0.0, 2.0], (E, A, 1)))
        assert v.sim.batch.step_launches() == want, (E, A)
        if want:
            v.step_async(np.tile([0.0, 2.0], (E, A, 1))); v.step_wait()
            assert v.sim.batch.step_launches() == 0, "step_async does not wait inside the call: the per-kernel form"
        v.sim.batch.close()
    img, res, origin = load_map_image("example_map")
    for mode in ("step", "step_device", "per_agent_noise", "profiling"):
        s = amd.BatchSim(num_envs=1, num_agents=2)
        s.set_map_image(img, res, origin)
        if mode == "per_agent_noise":
            s.set_noise_rng(None, per_agent_seeds=[1, 2])
        else:
            s.set_noise_rng(12345, 0.01)
        s.reset(bench_start_poses(1, 2))
        act = np.tile([0.0, 2.0], (2, 1))
        if mode == "profiling":
            s.profile_kernels(True)
        if mode == "step":
            s.step(act)
        elif mode == "step_device":
            d = s.device_array((2, 2)); d.upload(act); s.step_device(d)
        else:
            hb = s.host_block(("state", "collisions"))
            hb.actions[...] = act
            s.step_host(hb)
        assert s.step_launches() == 0, mode
        s.close()


@pytest.mark.parametrize("A", [2, 1])
def test_f110env_episodes_switch_between_the_step_forms_vs_oracle(amd, A):
    """tools/debug/f110env_soak.py: F110Env driven reset() / step() until done / reset() ... against the oracle at every step, the noise
    generator's row cache cut to 64 rows — episodes that outlive it leave the one-launch form for the per-kernel form in mid-episode,
    the next reset() brings it back: both forms and the switches between them must show up, and nothing may differ"""
    spec = importlib.util.spec_from_file_location("f110env_soak", os.path.join(ROOT, "tools", "debug", "f110env_soak.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    res = m.run(episodes=25, A=A, verbose=False)
    assert res, "mismatch against the oracle"
    forms, switches, _ = res
    assert forms[1] > 100 and forms[0] > 100 and switches >= 4, (forms, switches)


@pytest.mark.parametrize("A", [2, 1])
def test_f110env_long_episodes_on_the_default_noise_cache_vs_oracle(amd, A):
    """the same with the DEFAULT cache (rows generated ahead of need on their own stream, doublings from 256 on) and cars that crawl: episodes
    live through several doublings and re-allocations of the cache, later episodes replay its rows — every step against the oracle, all of
    them in the one-launch form"""
    spec = importlib.util.spec_from_file_location("f110env_soak", os.path.join(ROOT, "tools", "debug", "f110env_soak.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    res = m.run(episodes=5, A=A, cap=2600, rows=0, slow=True, verbose=False)
    assert res, "mismatch against the oracle"
    forms, switches, longest = res
    assert forms[0] == 0 and switches == 0 and longest > 1100, (forms, switches, longest)


NESTED = bool(os.environ.get("F110_NESTED_SUITE"))


@pytest.mark.parametrize("first", range(0, 200, 50))
def test_fuzz_host_block_step_vs_oracle(amd, first):
    """tools/debug/fuzz_host.py, seeds 0 .. 199: f110_step_host — every column of the page-locked block after every step — against the
    CPU oracle on random configurations weighted towards tiny batches, where the step is ONE launch (k_step_tiny); the library's choice of
    form is asserted per step"""
    spec = importlib.util.spec_from_file_location("fuzz_host", os.path.join(ROOT, "tools", "debug", "fuzz_host.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    bad = [sd for sd in range(first, first + (3 if NESTED else 50)) if not fz.run(sd, verbose=False)]
    assert not bad, bad


@pytest.mark.parametrize("E,A", [(1, 2), (1, 1), (48, 2)])
def test_noise_rows_generated_ahead_of_need_are_numpys_stream(amd, E, A):
    """the shared noise stream (laser_models.py:450-452, rng re-seeded at reset: base_classes.py:204) is cached row by row; the next
    doubling of the cache is generated AHEAD of need on a stream of its own and taken over by the step that reaches it (f110_hip.hip
    noise_start_ahead / noise_adopt), its memory continuing in a bigger block.  Cars at rest: scan(t) = min(range + row(t), opponent) with the SAME range
    every step, so row(t) must be NumPy's default_rng(12345).normal(0, 0.01, 1080) stream bit for bit — over 1400 steps (doublings at 256,
    512, 1024, two of them into new memory), a reset in the middle (rows start over, generation stays pending) and in both step forms
    (one launch for <= 4 agents, the per-kernel form for 96)"""
    B, T = 1080, 1400
    rows = np.random.default_rng(12345).normal(0., 0.01, size=(T + 2, B))
    poses = bench_start_poses(E, A)
    def quiet_scans(envs, agents):
        q = amd.BatchSim(num_envs=envs, num_agents=agents, num_beams=B)
        q.set_map(map_stem("example_map") + ".yaml", ".png")
        q.reset(poses); q.step(np.zeros((E * A, 2)))
        out = q.get("scans")["scans"].copy()
        q.close()
        return out
    free = quiet_scans(E * A, 1)                 # every car alone on the map
    seen = quiet_scans(E, A)                     # ... and with its opponent in view (ray_cast_agents lowers the beams that hit it)
    opp = np.where(seen < free, seen, np.inf)    # laser_models.py ray_cast: `if range < scan[i]: scan[i] = range`, applied AFTER the noise
    assert A == 1 or np.isfinite(opp).any()
    s = amd.BatchSim(num_envs=E, num_agents=A, num_beams=B)
    s.set_map(map_stem("example_map") + ".yaml", ".png"); s.set_noise_rng(12345, 0.01)
    s.reset(poses)
    hb = s.host_block(("scans", "state", "collisions", "in_collision"))
    hb.actions[...] = 0.0
    t_row = 0
    for t in range(T):
        if t == 700:
            s.reset(poses); t_row = 0        # RaceCar.reset re-seeds: the rows start over
        s.step_host(hb)
        assert s.step_launches() == (1 if E * A <= 4 else 0)
        got = hb.views["scans"]
        assert np.array_equal(got, np.minimum(free + rows[t_row][None, :], opp)), (t, t_row)
        t_row += 1
    s.close()


@pytest.mark.parametrize("groups", [2, 0, 1])
def test_noise_rows_ahead_of_need_under_env_blocks(amd, groups):
    """... and when the steps run as two env blocks on two streams (step_device back to back, step_groups 2 / automatic): the block
    streams must see the adopted rows (and the cache's new memory) exactly as the main stream does.  Cars at rest, bursts of device steps
    through the doublings at 256, 512 and 1024; after every burst the scans are min(range + row(t), opponent) with NumPy's row t"""
    E, A, B = 300, 2, 1080
    rows = np.random.default_rng(12345).normal(0., 0.01, size=(1400, B))
    poses = bench_start_poses(E, A)

    def quiet_scans(envs, agents):
        q = amd.BatchSim(num_envs=envs, num_agents=agents, num_beams=B)
        q.set_map(map_stem("example_map") + ".yaml", ".png")
        q.reset(poses); q.step(np.zeros((E * A, 2)))
        out = q.get("scans")["scans"].copy()
        q.close()
        return out
    free, seen = quiet_scans(E * A, 1), quiet_scans(E, A)
    opp = np.where(seen < free, seen, np.inf)
    s = amd.BatchSim(num_envs=E, num_agents=A, num_beams=B, step_groups=groups)
    s.set_map(map_stem("example_map") + ".yaml", ".png"); s.set_noise_rng(12345, 0.01)
    s.reset(poses)
    d_act = s.device_array((E * A, 2)); d_act.upload(np.zeros((E * A, 2)))
    t = 0
    for burst in (100, 27, 1, 130, 255, 3, 1, 500, 40, 250):
        for _ in range(burst):
            s.step_device(d_act)
        t += burst
        got = s.get("scans")["scans"]
        assert np.array_equal(got, np.minimum(free + rows[t - 1][None, :], opp)), (groups, t)
    s.close()


def test_duo_tail_equals_the_general_pair_body(amd):
    """k_step_tiny finishes one env of two cars in finalize_duo_tiny (three barriers, the scans copied into the host block by the idle
    wave while the role lanes work, shortened beams written through) — against finalize_pair_body in the same kernel (lab switch
    tiny_general_tail): the host block and the device state bit for bit over 600 fast, blind steps with wall hits, contacts, auto-resets"""
    outs = []
    for general in (0, 1):
        s = amd.BatchSim(num_envs=1, num_agents=2)
        s.set_map(map_stem("example_map") + ".yaml", ".png"); s.set_noise_rng(12345, 0.01)
        s.exp_set("tiny_general_tail", general)
        s.episode_init(0)
        poses = bench_start_poses(1, 2, gap_wp=3)
        s.episode_reset(poses)
        hb = s.host_block(("scans", "state", "agent_poses", "collisions", "collision_idx", "in_collision", "done", "lap_counts", "toggles"))
        rng = np.random.default_rng(3)
        rec = []
        for t in range(600):
            hb.actions[...] = np.stack([rng.uniform(-0.4, 0.4, 2), rng.uniform(3.0, 11.0, 2)], axis=1)
            s.step_host(hb, auto_reset=True)
            assert s.step_launches() == 1
            rec.append({k: v.copy() for k, v in hb.views.items()})
        rec.append(s.get("scans", "state", "collisions", "collision_idx", "in_collision", "agent_poses", "step_count"))
        outs.append(rec)
        s.close()
    assert sum(int(r["done"][0]) for r in outs[0][:-1]) > 2 and sum(float(r["collisions"].sum()) for r in outs[0][:-1]) > 2
    for t, (ra, rb) in enumerate(zip(*outs)):
        for k in ra:
            assert np.array_equal(ra[k], rb[k]), (t, k)


def test_host_logic_vec_env_through_both_observation_paths(amd):
    """Simulator.step brings the observation back through the page-locked block (one ABI call) up to 32 MB of scans and by plain copies above it
    (a staging block of hundreds of MB would cost more than the call it saves): 2100 envs x 2 cars sit ABOVE the line, the same envs in two
    halves below it — every observation key, `done` and the lap arrays of the big env equal those of the halves, over steps with wall hits"""
    E, A = 2100, 2
    kw = dict(map=map_stem("example_map"), map_ext=".png", num_agents=A)
    poses = bench_start_poses(E, A, gap_wp=4).reshape(E, A, 3)
    big = amd.F110VecEnv(E, **kw)
    halves = [amd.F110VecEnv(E // 2, **kw) for _ in range(2)]
    assert E * A * 1080 * 8 > (32 << 20) >= (E // 2) * A * 1080 * 8
    ob, _, db, ib = big.reset(poses)
    parts = [h.reset(poses[k * (E // 2):(k + 1) * (E // 2)]) for k, h in enumerate(halves)]
    rng = np.random.default_rng(4)
    hit = 0.0
    for t in range(90):
        if t % 15 == 1:
            act = np.stack([rng.uniform(-0.4, 0.4, (E, A)), rng.uniform(6.0, 14.0, (E, A))], axis=2)
        if t:
            ob, _, db, ib = big.step(act)
            parts = [h.step(act[k * (E // 2):(k + 1) * (E // 2)]) for k, h in enumerate(halves)]
        for key in ("scans", "poses_x", "poses_y", "poses_theta", "linear_vels_x", "linear_vels_y", "ang_vels_z", "collisions", "lap_times", "lap_counts"):
            assert np.array_equal(np.asarray(ob[key]), np.concatenate([np.asarray(p[0][key]) for p in parts])), (t, key)
        assert np.array_equal(db, np.concatenate([p[2] for p in parts])), t
        assert np.array_equal(ib["checkpoint_done"], np.concatenate([p[3]["checkpoint_done"] for p in parts])), t
        hit += float(np.asarray(ob["collisions"]).sum())
    assert hit > 0
    for e in [big] + halves:
        e.sim.batch.close()
