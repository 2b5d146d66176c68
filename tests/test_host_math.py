"""CPU check of the product's scalar math (f1tenth_gym_amd/csrc/f110_math.hpp).

The header is __host__ __device__; tests/host_harness compiles its HOST instantiation.  Here
that code is compared with the oracle and the golden vectors so ordering/arith mistakes are
caught without a GPU.  The GPU parity tests (-m gpu) prove the device instantiation.
"""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import orc
from _util import gold, oracle_map_dt, rel_err

HERE = os.path.dirname(os.path.abspath(__file__))
HDIR = os.path.join(HERE, "host_harness")
HLIB = os.path.join(HDIR, "libhost_harness.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.isfile("/opt/rocm/bin/hipcc"),
                                reason="hipcc needed to build the host harness")


@pytest.fixture(scope="module")
def hh():
    src = os.path.join(HDIR, "harness.hip")
    csrc = os.path.join(os.path.dirname(HERE), "f1tenth_gym_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in ("f110_math.hpp", "f110_rng.hpp", "f110_ziggurat_tables.hpp")]
    if not os.path.isfile(HLIB) or os.path.getmtime(HLIB) < max(os.path.getmtime(f) for f in deps):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC",
                               "-shared", src, "-o", HLIB], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    L = C.CDLL(HLIB)
    L.hh_get_range.restype = C.c_double
    return L


def d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def test_rhs_and_pid(hh):
    g = gold("dynamics")
    p, pp = d(g["params"])
    for x, u, fs, fk in zip(g["x"], g["u"], g["f_st"], g["f_ks"]):
        x_, xp = d(x); u_, up = d(u)
        f_st = np.empty(7); f_ks = np.empty(5)
        hh.hh_rhs(xp, up, pp, f_st.ctypes.data_as(_dp), f_ks.ctypes.data_as(_dp))
        # bit-equal to the oracle (same libm); NumPy's tan differs from libm by 1 ulp on a few inputs
        assert np.array_equal(f_st, orc.vehicle_dynamics_st(x, u, p)) and np.array_equal(f_ks, orc.vehicle_dynamics_ks(x[:5], u, p))
        assert rel_err(f_st, fs) < 1e-12 and rel_err(f_ks, fk) < 1e-12
    for r, o in zip(g["pid_in"], g["pid_out"]):
        r_, rp = d(r); out = np.empty(2)
        hh.hh_pid(rp, pp, out.ctypes.data_as(_dp))
        assert np.array_equal(out, o)


@pytest.mark.parametrize("name,integ,ld", [("rk4", 1, 0.0), ("euler", 2, 0.0), ("rk4_lidar", 1, 0.275)])
def test_advance_vehicle(hh, name, integ, ld):
    g = gold("update_pose")
    p, pp = d(g["params"])
    for i in range(g[name + "_state0"].shape[0]):
        st = np.array(g[name + "_state0"][i]); buf = np.zeros(2)
        c0 = int(g[name + "_cnt0"][i])
        buf[:c0] = g[name + "_buf0"][i, :c0]
        cnt = C.c_int(c0); sp = np.empty(3)
        hh.hh_advance(st.ctypes.data_as(_dp), buf.ctypes.data_as(_dp), C.byref(cnt), C.c_double(g[name + "_action"][i, 0]),
                      C.c_double(g[name + "_action"][i, 1]), pp, C.c_double(0.01), integ, C.c_double(ld),
                      sp.ctypes.data_as(_dp))
        o_st, o_sb, o_cnt, o_sp = orc.update_pose(g[name + "_state0"][i], buf if False else np.concatenate([g[name + "_buf0"][i, :c0], np.zeros(2 - c0)]),
                                                  c0, g[name + "_action"][i, 0], g[name + "_action"][i, 1], p, 0.01, integ, ld)
        assert np.array_equal(st, o_st) and np.array_equal(sp, o_sp)
        assert rel_err(st, g[name + "_state1"][i]) < 1e-12
        assert cnt.value == g[name + "_cnt1"][i]
        assert np.array_equal(buf[:cnt.value], g[name + "_buf1"][i, :cnt.value])
        assert rel_err(sp, g[name + "_scan_pose"][i]) < 1e-12


@pytest.mark.parametrize("integ", [1, 2])
def test_two_wave_integration_is_the_one_wave_integration(hh, integ):
    """k_integrate_duo's decomposition, on the host: a second walker takes (steer, v) through the RK4 stages alone
    (low_speed_trig_ahead) and leaves the low-speed branch's tan / cos in a table; the integration takes them from
    there.  Bit-identical to advance_vehicle for random states on both sides of |v| = 0.5 (and crossing it inside a
    step), every delay-buffer fill, steering at its limits, both integrators — and a stage consumes a table entry
    exactly when the walker produced one."""
    g = gold("update_pose")
    p, pp = d(g["params"])
    rng = np.random.default_rng(77 + integ)
    n_low = n_high = n_cross = 0
    for i in range(6000):
        st0 = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-0.45, 0.45), 0.0, rng.uniform(-7, 7), rng.uniform(-2, 2), rng.uniform(-0.3, 0.3)])
        kind = i % 4
        st0[3] = (rng.uniform(-0.6, 0.6), rng.uniform(0.4, 0.6) * rng.choice([-1, 1]), rng.uniform(-6, 12), 0.0)[kind]
        if i % 7 == 0:
            st0[2] = rng.choice([-0.4189, 0.4189])          # at the steering limits (the rate constraint's 0 branch)
        c0 = int(rng.integers(0, 3))
        buf0 = np.concatenate([rng.uniform(-0.4, 0.4, c0), np.zeros(2 - c0)])
        steer, speed = rng.uniform(-0.5, 0.5), rng.uniform(-3, 15)
        outs = []
        for fn in ("hh_advance", "hh_advance_duo"):
            st = st0.copy(); buf = buf0.copy(); cnt = C.c_int(c0); sp = np.empty(3); which = np.zeros(4, dtype=np.intc)
            args = [st.ctypes.data_as(_dp), buf.ctypes.data_as(_dp), C.byref(cnt), C.c_double(steer), C.c_double(speed), pp, C.c_double(0.01), integ,
                    C.c_double(0.275 if i % 5 == 0 else 0.0), sp.ctypes.data_as(_dp)]
            if fn == "hh_advance_duo":
                args.append(which.ctypes.data_as(_ip))
            getattr(hh, fn)(*args)
            outs.append((st, buf, cnt.value, sp, which))
        (s1, b1, c1, p1, _), (s2, b2, c2, p2, which) = outs
        assert np.array_equal(s1, s2) and np.array_equal(b1, b2) and c1 == c2 and np.array_equal(p1, p2), (i, st0)
        stages = 4 if integ == 1 else 1
        assert all(w in (0, 3) for w in which[:stages]) and all(w == 0 for w in which[stages:]), which   # produced <=> consumed
        low = [w == 3 for w in which[:stages]]
        n_low += all(low); n_high += not any(low); n_cross += any(low) and not all(low)
    assert n_low > 500 and n_high > 500 and (n_cross > 20 or integ == 2), (n_low, n_high, n_cross)


def test_fan_integration_is_the_one_wave_integration(hh):
    """k_integrate_fan's decomposition, on the host (round 4): per stage one role computes the low-speed branch's f4 / f5
    (fan_low) or the single-track branch's coefficients (fan_dyn) from the RAW state, the main chain (fan_main) advances
    steer / velocity / yaw / yaw rate / slip taking them from tables, and the position derivatives come from fan_pos
    afterwards.  Bit-identical to advance_vehicle (RK4) on both sides of |v| = 0.5, crossing it inside a step, every
    delay-buffer fill, steering at its limits, lidar on and off the axle, yaw wrap both ways"""
    g = gold("update_pose")
    p, pp = d(g["params"])
    rng = np.random.default_rng(404)
    n_low = n_high = n_cross = 0
    for i in range(8000):
        st0 = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-0.45, 0.45), 0.0, rng.uniform(-7, 7), rng.uniform(-2, 2), rng.uniform(-0.3, 0.3)])
        kind = i % 4
        st0[3] = (rng.uniform(-0.6, 0.6), rng.uniform(0.4, 0.6) * rng.choice([-1, 1]), rng.uniform(-6, 12), 0.0)[kind]
        if i % 7 == 0:
            st0[2] = rng.choice([-0.4189, 0.4189])
        if i % 11 == 0:
            st0[4] = rng.choice([-1e-3, 2 * np.pi + 1e-3, 0.0, 2 * np.pi])     # around the yaw wrap
        if i % 13 == 0:
            st0[0] = -0.0                                                          # the on-axle shortcut's exception
        c0 = int(rng.integers(0, 3))
        buf0 = np.concatenate([rng.uniform(-0.4, 0.4, c0), np.zeros(2 - c0)])
        steer, speed = rng.uniform(-0.5, 0.5), rng.uniform(-3, 15)
        ld = 0.275 if i % 5 == 0 else 0.0
        outs = []
        for fn in ("hh_advance", "hh_advance_fan"):
            st = st0.copy(); buf = buf0.copy(); cnt = C.c_int(c0); sp = np.empty(3); which = np.zeros(4, dtype=np.intc)
            if fn == "hh_advance":
                hh.hh_advance(st.ctypes.data_as(_dp), buf.ctypes.data_as(_dp), C.byref(cnt), C.c_double(steer), C.c_double(speed), pp, C.c_double(0.01), 1,
                              C.c_double(ld), sp.ctypes.data_as(_dp))
            else:
                hh.hh_advance_fan(st.ctypes.data_as(_dp), buf.ctypes.data_as(_dp), C.byref(cnt), C.c_double(steer), C.c_double(speed), pp, C.c_double(0.01),
                                  C.c_double(ld), sp.ctypes.data_as(_dp), which.ctypes.data_as(_ip))
            outs.append((st, buf, cnt.value, sp, which))
        (s1, b1, c1, p1, _), (s2, b2, c2, p2, which) = outs
        same = lambda a, b: np.array_equal(a.view(np.uint64), b.view(np.uint64))     # bit patterns (incl. the sign of zero)
        assert same(s1, s2) and same(b1, b2) and c1 == c2 and same(p1, p2), (i, st0, s1, s2)
        low = [w == 1 for w in which]
        n_low += all(low); n_high += not any(low); n_cross += any(low) and not all(low)
    assert n_low > 500 and n_high > 500 and n_cross > 20, (n_low, n_high, n_cross)


def _hh_scan(hh, layout, dt, res, origin, sines, cosines, B, fov, pose, theta_dis=2000):
    dt_, dtp = d(dt); s_, sp = d(sines); c_, cp = d(cosines); pose_, pp = d(pose)
    ranges = np.empty(B); hits = np.empty((B, 2), dtype=np.intc); idx = np.empty(B, dtype=np.intc)
    lk = C.c_longlong(0)
    hh.hh_scan(layout, dtp, dt.shape[0], dt.shape[1], C.c_double(res), C.c_double(origin[0]), C.c_double(origin[1]),
               C.c_double(np.cos(origin[2])), C.c_double(np.sin(origin[2])), sp, cp, theta_dis, B, C.c_double(fov),
               C.c_double(1e-4), C.c_double(30.0), pp, ranges.ctypes.data_as(_dp), hits.ctypes.data_as(_ip),
               idx.ctypes.data_as(_ip), C.byref(lk))
    return ranges, hits, idx, lk.value


@pytest.mark.parametrize("fixture,mapname,beams,fov", [
    ("scan_example_map", "example_map", 1080, 4.7), ("scan_berlin", "berlin", 1080, 4.7),
    ("scan_example_map_4096", "example_map", 4096, 4.7), ("scan_example_map_271", "example_map", 271, 6.0)])
@pytest.mark.parametrize("layout", [0, 1, 2, 3])
def test_scan_matches_golden(hh, fixture, mapname, beams, fov, layout):
    g = gold(fixture)
    dt, res, origin = oracle_map_dt(mapname)
    so = orc.ScanOracle(beams, fov)
    for k, pose in enumerate(g["poses"]):
        ranges, hits, idx, lk = _hh_scan(hh, layout, dt, res, origin, so.sines, so.cosines, beams, fov, pose)
        assert np.array_equal(idx, g["dir_idx"][k])
        assert np.array_equal(hits, g["hit_rc"][k])
        assert np.array_equal(ranges, g["scans"][k])
        assert lk == g["lookups"][k]


def test_scan_generic_path_and_rotated_origin(hh):
    """non-power-of-two resolution + rotated origin take the guarded-division / rotation code;
    compare with the oracle on a synthetic transform of example_map's table."""
    dt, res, origin = oracle_map_dt("example_map")
    sub = np.ascontiguousarray(dt[600:1000, 900:1300])
    so = orc.ScanOracle(1080, 4.7)
    rng = np.random.default_rng(11)
    for res2, org in [(0.05, [-3.0, -4.0, 0.3]), (0.0625, [1.0, 2.0, -1.1]), (0.07, [0.0, 0.0, 0.0])]:
        so.set_map_dt(sub * (res2 / res), res2, org)
        c, s = np.cos(org[2]), np.sin(org[2])
        for _ in range(6):
            u, v = rng.uniform(5, 15, 2) * res2 / 0.05
            pose = [org[0] + c * u - s * v, org[1] + s * u + c * v, rng.uniform(0, 6.28)]
            ref, ref_hits = so.scan(pose, want_hits=True)
            for layout in (0, 3, 203, 1603):
                ranges, hits, idx, lk = _hh_scan(hh, layout, so.dt, res2, org, so.sines, so.cosines, 1080, 4.7, pose)
                assert np.array_equal(hits, ref_hits)
                assert np.array_equal(ranges, ref)
                assert lk == so.last_lookups


def _padded_stats(hh):
    out = (C.c_longlong * 3)()
    hh.hh_padded_stats(out)
    return dict(fast=out[0], guard=out[1], far=out[2])


def test_padded_layout_guard_band_and_far_poses(hh):
    """PADDED layout (border of out-of-bounds cells + fixed-point cell addressing): bit-equal to
    the oracle where the cheap decision is taken, where a sample falls in the guard band (rays
    running along cell boundaries) and where the lidar is too far off the map for the border."""
    so = orc.ScanOracle(1080, 4.7)
    rng = np.random.default_rng(21)
    _padded_stats(hh)
    for mapname in ("berlin", "example_map", "skirk"):
        dt, res, origin = oracle_map_dt(mapname)
        so.set_map_dt(dt, res, origin)
        H, W = dt.shape
        free = np.argwhere(dt > 0.3)
        poses = []
        for r, c in free[rng.choice(len(free), 6, replace=False)]:
            # lidar exactly on a cell corner / edge, headings along the axes: samples land on
            # cell boundaries over and over
            poses.append([origin[0] + c * res, origin[1] + r * res, rng.choice([0.0, np.pi / 2, np.pi, -np.pi / 2]) + 4.7 / 2 * 0])
            poses.append([origin[0] + c * res, origin[1] + (r + 0.5) * res, 0.0])
            # beam 0 takes table direction 0 = (1, 0) exactly: it runs along the cell boundary y = const
            poses.append([origin[0] + (c + 0.25) * res, origin[1] + r * res, 4.7 / 2 + 1e-5])
            poses.append([origin[0] + (c + rng.uniform()) * res, origin[1] + (r + rng.uniform()) * res, rng.uniform(-7, 7)])
        # on the border of the map, just outside, far outside, absurd
        poses += [[origin[0], origin[1], 0.3], [origin[0] - 1.0, origin[1] + H * res / 2, 0.0],
                  [origin[0] + W * res + 2.5, origin[1] + H * res + 2.5, 3.9], [origin[0] - 40.0, origin[1] - 40.0, 0.8],
                  [origin[0] + W * res / 2, origin[1] + H * res + 29.0, -1.6], [1e9, -1e9, 1.0], [1e300, 0.0, 0.0]]
        for pose in poses:
            ref, ref_hits = so.scan(pose, want_hits=True)
            ranges, hits, idx, lk = _hh_scan(hh, 3, dt, res, origin, so.sines, so.cosines, 1080, 4.7, pose)
            assert np.array_equal(hits, ref_hits), (mapname, pose)
            assert np.array_equal(ranges, ref), (mapname, pose)
            assert lk == so.last_lookups
    st = _padded_stats(hh)
    # the rays that run along a cell boundary are re-marched exactly (and only a handful of others);
    # the far-off lidars never use the border
    assert st["fast"] > 20000 and 18 <= st["guard"] < 1e-3 * st["fast"] and st["far"] >= 3 * 2 * 1080, st


def test_dir_index_exact_replay(hh):
    """closed-form beam index == sequential replay (guard=2 forces the replay on every beam)
    == oracle, for many headings incl. wrap boundaries."""
    so = orc.ScanOracle(1080, 4.7)
    rng = np.random.default_rng(12)
    thetas = list(rng.uniform(-10, 10, 60)) + [0.0, 2.35, 2.35 + 1e-12, np.pi, 2 * np.pi, -2.35, 4.7 / 2]
    for B, fov in [(1080, 4.7), (4096, 4.7), (271, 6.0)]:
        so = orc.ScanOracle(B, fov)
        for th in thetas:
            ref = so.beam_dir_indices(th)
            a = np.empty(B, dtype=np.intc); b = np.empty(B, dtype=np.intc)
            hh.hh_dir_index(2000, B, C.c_double(fov), C.c_double(th), C.c_double(1e-8), a.ctypes.data_as(_ip))
            hh.hh_dir_index(2000, B, C.c_double(fov), C.c_double(th), C.c_double(2.0), b.ctypes.data_as(_ip))
            assert np.array_equal(a, ref) and np.array_equal(b, ref)


def test_gjk_vertices_ttc(hh):
    g = gold("collision")
    L, W = g["length"][0], g["width"][0]
    for pa, va in zip(g["pose_a"][:200], g["vert_a"][:200]):
        p_, pp = d(pa); v = np.empty(8)
        hh.hh_vertices(pp, C.c_double(L), C.c_double(W), v.ctypes.data_as(_dp))
        assert np.array_equal(v.reshape(4, 2), orc.get_vertices(pa, L, W))
        assert rel_err(v.reshape(4, 2), va) < 1e-12
    flags = []
    for a, b in zip(g["vert_a"], g["vert_b"]):
        a_, ap = d(a); b_, bp = d(b)
        flags.append(hh.hh_gjk(ap, bp))
    assert np.array_equal(flags, g["flags"])
    np.random.seed(1234)   # collision_models.py:274,306-311
    v1 = np.asarray([[4, 11.], [5, 5], [9, 9], [10, 10]])
    for _ in range(300):
        a_, ap = d(v1 + np.random.normal(size=v1.shape) / 100.); b_, bp = d(v1 + np.random.normal(size=v1.shape) / 100.)
        assert hh.hh_gjk(ap, bp) == 1
    t = gold("ttc")
    co, cp = d(t["cosines"]); sd, sp = d(t["side_distances"])
    out = []
    for s, v in zip(t["scans"], t["vels"]):
        s_, spn = d(s)
        out.append(hh.hh_ttc(spn, 1080, C.c_double(v), cp, sp, C.c_double(0.005)))
    assert np.array_equal(out, t["flags"])


def test_raycast_and_get_range(hh):
    g = gold("raycast")
    sa, sap = d(g["scan_angles"])
    for i in range(g["ego"].shape[0]):
        e_, ep = d(g["ego"][i]); v_, vp = d(g["vertices"][i])
        scan = np.full(1080, g["base"][0]); mm = np.empty(4, dtype=np.intc)
        hh.hh_raycast(ep, vp, sap, 1080, scan.ctypes.data_as(_dp), mm.ctypes.data_as(_ip))
        assert tuple(mm[:2]) == (g["min_ind"][i], g["max_ind"][i])
        ref = orc.ray_cast(g["ego"][i], np.full(1080, g["base"][0]), g["scan_angles"], g["vertices"][i])
        assert np.array_equal(scan, ref)
        assert np.array_equal(scan != g["base"][0], g["scans"][i] != g["base"][0])
    out = []
    for r in g["get_range_in"]:
        r_, rp = d(r)
        out.append(hh.hh_get_range(rp))
    out = np.array(out); ref = g["get_range_out"]
    assert np.array_equal(np.isinf(out), np.isinf(ref))
    fin = ~np.isinf(ref)
    assert rel_err(out[fin], ref[fin]) < 1e-9


def test_get_range_division_free_decisions_equal_the_reference_expression(hh):
    """edge_range decides hit / miss on the numerators and divides only for an edge that is hit (round 4); it must
    return exactly what get_range's own expression returns (laser_models.py:249-280: d1 = n1 / denom, d2 = n2 / denom,
    hit <=> d1 >= 0 and 0 <= d2 <= 1) — random boxes, rays through corners (d2 at 0 / 1 to the last bit), rays parallel
    to an edge, the lidar on an edge's line, tiny and huge operands"""
    import math
    rng = np.random.default_rng(77)

    def ref(row):
        ox, oy, _, bt, vax, vay, vbx, vby = row
        v3x, v3y = math.cos(bt + math.pi / 2.), math.sin(bt + math.pi / 2.)
        v1x, v1y = ox - vax, oy - vay
        v2x, v2y = vbx - vax, vby - vay
        denom = v2x * v3x + v2y * v3y
        if abs(denom) > 0.0:
            n1 = v2x * v1y - v2y * v1x
            n2 = v1x * v3x + v1y * v3y
            with np.errstate(all="ignore"):
                d1 = float(np.float64(n1) / np.float64(denom)); d2 = float(np.float64(n2) / np.float64(denom))
            return d1 if (d1 >= 0.0 and d2 >= 0.0 and d2 <= 1.0) else math.inf
        return None   # collinear branch: not what this test is about

    rows = []
    for k in range(30000):
        o = rng.uniform(-3, 3, 2); bt = rng.uniform(-4, 4)
        va = rng.uniform(-3, 3, 2); vb = va + rng.uniform(-0.7, 0.7, 2)
        kind = k % 6
        if kind == 1:      # the ray aimed exactly at an end point (d2 == 0 or 1 up to rounding)
            tgt = va if k % 12 == 1 else vb
            bt = math.atan2(tgt[1] - o[1], tgt[0] - o[0])
        elif kind == 2:    # ... and one ulp-ish beside it
            tgt = vb
            bt = math.atan2(tgt[1] - o[1], tgt[0] - o[0]) + rng.choice([-1, 1]) * rng.choice([1e-16, 3e-16, 1e-15, 1e-13, 1e-12, 1e-11])
        elif kind == 3:    # nearly parallel to the edge
            bt = math.atan2(vb[1] - va[1], vb[0] - va[0]) + rng.choice([-1, 1]) * rng.choice([0.0, 1e-17, 1e-15, 1e-12, 1e-9])
        elif kind == 4:    # the lidar on the edge's line / on a vertex
            tt = rng.uniform(-1, 2)
            o = va + tt * (vb - va) if k % 3 else va.copy()
        elif kind == 5:    # scaled operands: tiny, huge
            sc = 10.0 ** rng.choice([-160, -120, -60, 60, 120, 160])
            o, va, vb = o * sc, va * sc, vb * sc
        rows.append([o[0], o[1], 0.0, bt, va[0], va[1], vb[0], vb[1]])
    n_hit = n_band = 0
    for row in rows:
        want = ref(row)
        if want is None:
            continue
        r_, rp = d(np.array(row))
        got = hh.hh_get_range(rp)
        assert (got == want) or (math.isnan(got) and math.isnan(want)), (row, got, want)
        n_hit += int(math.isfinite(want))
    assert n_hit > 3000


def test_disc_cull_is_result_preserving(hh):
    """the product only evaluates window beams whose ray can touch the opponent's circumscribed
    disc; the oracle evaluates the whole [min_ind, max_ind] window like the reference.  Results
    must be identical on random placements, incl. opponents behind / overlapping / far away."""
    sa, co, sd = orc.build_beam_tables(1080, 4.7, 0.31, 0.15875, 0.17145)
    sa_, sap = d(sa)
    rng = np.random.default_rng(21)
    culled_total = ref_total = 0
    for i in range(1500):
        ego = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(0, 2 * np.pi) if i % 7 else 0.0])
        if i % 11 == 0:      # diverged yaw (a reversing car can blow the yaw rate up): huge headings
            ego[2] = rng.choice([1.16059365e+15, -3.3e12, 7.7e8, 1e300])
        dist = rng.uniform(0.0, 0.5) if i % 10 == 0 else rng.uniform(0.3, 12.0)
        bearing = rng.uniform(-np.pi, np.pi) if i % 3 else np.pi + rng.uniform(-0.5, 0.5)
        opp = np.array([ego[0] + dist * np.cos(ego[2] + bearing), ego[1] + dist * np.sin(ego[2] + bearing), rng.uniform(0, 2 * np.pi)])
        v = orc.get_vertices(opp, 0.58, 0.31)
        base = rng.uniform(0.2, 15.0, 1080)
        ref = orc.ray_cast(ego, base, sa, v)
        e_, ep = d(ego); v_, vp = d(v)
        scan = base.copy(); mm = np.empty(4, dtype=np.intc)
        hh.hh_raycast(ep, vp, sap, 1080, scan.ctypes.data_as(_dp), mm.ctypes.data_as(_ip))
        assert tuple(mm[:2]) == orc.get_blocked_view_indices(ego, v, sa)
        assert np.array_equal(scan, ref), i
        ref_total += mm[1] - mm[0] + 1
        culled_total += max(0, mm[3] - mm[2] + 1)
    assert culled_total < 0.7 * ref_total     # the cull actually removes work


def test_ttc_predicate_guard_band(hh):
    """the division-free iTTC predicate must agree with the reference expression everywhere,
    in particular within a few ulp of the threshold and at the 0 / inf / nan corner cases"""
    rng = np.random.default_rng(31)
    B = 64
    co = np.cos(np.linspace(-2.35, 2.35, B)); sd = rng.uniform(0.1, 0.4, B)
    thresh = 0.005
    co_, cp = d(co); sd_, sp = d(sd)
    cases = []
    for _ in range(400):
        v = rng.uniform(-6, 8)
        scan = rng.uniform(0.0, 5.0, B)
        j = rng.integers(B)
        den = v * co[j]
        for delta in (0.0, 1e-16, -1e-16, 3e-16, -3e-16, 1e-13, -1e-13, 1e-11, -1e-11, 1e-9, -1e-9):
            s2 = scan.copy(); s2[:] = 5.0 + sd        # everything else far away
            s2[j] = sd[j] + thresh * den * (1.0 + delta)
            cases.append((s2, v))
    cases += [(np.full(B, 1.0), 0.0), (sd.copy(), 3.0), (sd.copy(), -3.0), (np.full(B, np.inf), 2.0),
              (np.full(B, np.nan), 2.0), (sd * 0.5, 2.0), (sd * 0.5, -2.0)]
    n_hit = 0
    for scan, v in cases:
        s_, spn = d(scan)
        got = hh.hh_ttc(spn, B, C.c_double(v), cp, sp, C.c_double(thresh))
        ref = int(orc.check_ttc(scan, v, co, sd, thresh))
        assert got == ref, (v, scan[:4])
        n_hit += ref
    assert 100 < n_hit < len(cases) - 100


def test_pure_pursuit_planner(hh):
    """examples/waypoint_follow.py planner: host instantiation of the device code == oracle (bit for
    bit: same libm) == the reference's outputs (golden; NumPy's dot / arctan differ by <= 1 ulp)"""
    g = gold("planner")
    wp, wpp = d(g["waypoints"]); M = wp.shape[0]
    L, vg, wb = float(g["tlad"][0]), float(g["vgain"][0]), float(g["wheelbase"][0])
    rng = np.random.default_rng(31)
    extra = np.stack([rng.uniform(-60, 20, 300), rng.uniform(-30, 30, 300), rng.uniform(-7, 7, 300)], axis=1)
    poses = np.concatenate([g["poses"], extra])
    for k, pose in enumerate(poses):
        p_, pp = d(pose)
        act = np.empty(2); ndt = np.empty(2); ni = C.c_int(0); goal = C.c_int(0)
        hh.hh_pure_pursuit(wpp, M, pp, C.c_double(L), C.c_double(vg), C.c_double(wb), C.c_double(20.0), act.ctypes.data_as(_dp),
                           C.byref(ni), ndt.ctypes.data_as(_dp), C.byref(goal))
        oi, od, ot = orc.nearest_on_trajectory(wp, pose[0], pose[1])
        assert ni.value == oi and ndt[0] == od and ndt[1] == ot
        if od < L:
            assert goal.value == orc.first_point_on_circle(wp, pose[0], pose[1], L, oi + ot)
        assert np.array_equal(act, orc.pure_pursuit_plan(wp, pose, L, vg, wb)), (k, pose)
        if k < len(g["poses"]):
            assert ni.value == int(g["nearest"][k, 0]) and abs(ndt[0] - g["nearest"][k, 1]) < 1e-12 and abs(ndt[1] - g["nearest"][k, 2]) < 1e-12
            li = int(g["lookahead_index"][k])
            if li >= -1:
                assert goal.value == (li if li >= 0 else M + li)
            assert np.max(np.abs(act - g["actions"][k])) < 1e-14



def test_per_agent_table_decomposition_of_the_opponent_window(hh):
    """what k_finalize_multi / k_finalize_multi_tiled keep per AGENT (cos / sin of the snapshot heading, the ray-cast heading's
    atan2) and then combine per record must be the straightforward computation bit for bit: box_vertices_cs(cos_sin(th)) ==
    box_vertices(th), and corner beam indices + disc cull from the cached pieces == opponent_beam_window — opponents
    everywhere around the ego (in front, exactly behind: the +-pi wrap, touching, far), headings of every size"""
    rng = np.random.default_rng(2024)
    B = 1080
    fov = 4.7
    angles = np.ascontiguousarray(np.linspace(-fov / 2., fov / 2., B))
    inc = fov / (B - 1)
    ap = angles.ctypes.data_as(_dp)
    n_empty = n_full = 0
    for k in range(6000):
        eth = rng.uniform(-np.pi, np.pi) * rng.choice([1.0, 1.0, 7.0, 1e3])
        dist = rng.choice([0.3, 0.6, 1.5, 5.0, 25.0]) * rng.uniform(0.8, 1.2)
        bearing = rng.uniform(-np.pi, np.pi) if k % 5 else np.pi * rng.choice([-1.0, 1.0]) + rng.uniform(-1e-3, 1e-3)   # every fifth: right behind
        ego = np.array([rng.uniform(-20, 20), rng.uniform(-20, 20), eth])
        opp = np.array([ego[0] + dist * np.cos(eth + bearing), ego[1] + dist * np.sin(eth + bearing), rng.uniform(-np.pi, np.pi) * rng.choice([1.0, 50.0])])
        v1, v2 = np.empty(8), np.empty(8)
        w1, w2 = np.zeros(4, dtype=np.int32), np.zeros(4, dtype=np.int32)
        hh.hh_box_and_window(ego.ctypes.data_as(_dp), opp.ctypes.data_as(_dp), C.c_double(0.58), C.c_double(0.31), ap, B, C.c_double(inc),
                             v1.ctypes.data_as(_dp), v2.ctypes.data_as(_dp), w1.ctypes.data_as(_ip), w2.ctypes.data_as(_ip))
        assert np.array_equal(v1, v2), k
        assert np.array_equal(w1, w2), (k, w1, w2)
        n_empty += int(w1[3] < w1[2]); n_full += int(w1[3] - w1[2] > B // 2)
    assert n_empty > 0 and n_full > 0   # both ends of the window logic were exercised
