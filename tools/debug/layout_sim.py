"""Pre-registration for round 6's one scan experiment: how many distinct 128-byte lines a 64-ray gather of
k_scan_rays_agent touches under three layouts of the PADDED table, on the steady regime's rays (the same rays
tools/debug/compaction_sim.py prices: oracle rollout of the bench workload, then every ray's sample sequence in NumPy).

  row-major   16 cells of one row per line (the product)                 line = (r * PW + c) >> 4
  4x4 tiles   one tile per line (round 5's lab variant, measured -2.6 %)  line = (r >> 2, c >> 2)
  row pairs   2 rows x 8 cells per line (this round's experiment)        line = (r >> 1, c >> 3)

    python tools/debug/layout_sim.py [envs=256] [steps=320]
CPU only (the oracle marches nothing here; NumPy does)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import oracle_map_dt, bench_start_poses  # noqa: E402
from oracle import orc  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 320
A, B = 2, 1080
dt, res, origin = oracle_map_dt("example_map")
poses = bench_start_poses(E, A)
rng = np.random.default_rng(1000)
sets = np.stack([np.stack([rng.uniform(-0.2, 0.2, E * A), rng.uniform(2.0, 6.0, E * A)], axis=1) for _ in range(T // 20)])
sim = orc.SimOracle(E, A); sim.set_map_dt(dt, res, origin)
sim.set_noise(np.random.default_rng(12345).normal(0., .01, size=(T + 2, 1080)))
sim.reset(poses)
t0 = time.time(); sim.rollout(sets, T, 20, poses, True, 8)
print("# oracle rollout: %d envs x %d agents, %d steps (%.1f s)" % (E, A, T, time.time() - t0))
st = sim.state.copy()
theta_dis, fov = 2000, 4.7
sines = np.sin(np.linspace(0, 2 * np.pi, theta_dis)); cosines = np.cos(np.linspace(0, 2 * np.pi, theta_dis))
inc = theta_dis * (fov / (B - 1)) / (2 * np.pi)
N = E * A
TPA = (B + 63) // 64
ti = theta_dis * (st[:, 4] - fov / 2.) / (2 * np.pi); ti = np.fmod(ti, theta_dis); ti[ti < 0] += theta_dis
idx = np.empty((N, B), dtype=np.int64)
cur = ti.copy()
for b in range(B):
    idx[:, b] = cur.astype(np.int64)
    cur = cur + inc; cur[cur >= theta_dis] -= theta_dis
idx = idx.reshape(-1); c = cosines[idx]; s = sines[idx]
H, W = dt.shape
PADB = int(np.ceil(30.0 / res)) + 66
PW = W + 2 * PADB
LAYOUTS = {"row-major 1x16": lambda r, cc: (r * PW + cc) >> 4,
           "tiles 4x4": lambda r, cc: (r >> 2) * (1 << 20) + (cc >> 2),
           "row pairs 2x8": lambda r, cc: (r >> 1) * (1 << 20) + (cc >> 3),
           "row quads 4x4 (= tiles)": None, "rows 8x2": lambda r, cc: (r >> 3) * (1 << 20) + (cc >> 1)}
LAYOUTS = {k: v for k, v in LAYOUTS.items() if v is not None}


def lookup(x, y):
    xt = x - origin[0]; yt = y - origin[1]
    oob = (xt < 0) | (xt >= W * res) | (yt < 0) | (yt >= H * res)
    cc = (xt / res).astype(np.int64); rr = (yt / res).astype(np.int64)
    pr, pc = rr + PADB, cc + PADB
    cc[oob] = -1; rr[oob] = -1
    return dt[rr, cc], pr, pc


xs = np.repeat(st[:, 0], B); ys = np.repeat(st[:, 1], B)
agent = np.repeat(np.arange(N), B); beam = np.tile(np.arange(B), N)
task = agent * TPA + beam // 64
d, pr0, pc0 = lookup(xs, ys)
total = d.copy()
active = (d > 1e-4) & (total <= 30.0)
x = xs.copy(); y = ys.copy()
prev = {k: f(pr0, pc0) for k, f in LAYOUTS.items()}     # the line of the ray's previous sample
stats = {k: dict(gathers=0, lines=0, same=0, samples=0) for k in LAYOUTS}
it = 0
while active.any():
    ia = np.nonzero(active)[0]
    x[ia] += d[ia] * c[ia]; y[ia] += d[ia] * s[ia]
    dn, pr, pc = lookup(x[ia], y[ia]); d[ia] = dn; total[ia] += dn
    for k, f in LAYOUTS.items():
        line = f(pr, pc)
        key = task[ia] * (1 << 42) + line
        uk = np.unique(key)
        stats[k]["gathers"] += np.unique(uk >> 42).size
        stats[k]["lines"] += uk.size
        stats[k]["same"] += int((line == prev[k][ia]).sum())
        stats[k]["samples"] += ia.size
        prev[k][ia] = line
    active[ia] = (dn > 1e-4) & (total[ia] <= 30.0)
    it += 1
print("# rays %d, march iterations %d, wave-level gathers %d (%.2f per task), samples per gather (active lanes) %.1f"
      % (x.size, it, stats["row-major 1x16"]["gathers"], stats["row-major 1x16"]["gathers"] / float(N * TPA),
         stats["row-major 1x16"]["samples"] / float(stats["row-major 1x16"]["gathers"])))
base = stats["row-major 1x16"]["lines"] / float(stats["row-major 1x16"]["gathers"])
print("%-26s %14s %10s %22s" % ("layout", "lines / gather", "vs today", "samples on the ray's previous line"))
for k, v in stats.items():
    lpg = v["lines"] / float(v["gathers"])
    print("%-26s %14.2f %10.3f %21.1f %%" % (k, lpg, lpg / base, 100.0 * v["same"] / v["samples"]))
