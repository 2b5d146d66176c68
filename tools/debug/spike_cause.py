"""Why does the 4096-agent scan take 100-170 us in a few steps out of 70 (profiles/r04_scan_series_4096.txt)?  On the CPU:
the bench workload rolled to its steady regime with the oracle, then step by step the per-ray sample counts (NumPy
restatement of trace_ray).  For every ray above LONG samples: was its 64-beam task on the longest-first list (its maximum
in the PREVIOUS step above the list threshold 96)?  was the agent re-seated in the previous step (no history)?  was a
NEIGHBOURING task listed?
usage: python tools/debug/spike_cause.py [envs=2048] [steps=24]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from _util import oracle_map_dt, bench_start_poses
from oracle import orc
E = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 24
A, B, T0, LONG, THR = 2, 1080, 300, 250, 96
dt, res, origin = oracle_map_dt("example_map")
poses = bench_start_poses(E, A)
rng = np.random.default_rng(1000)
sets = np.stack([np.stack([rng.uniform(-0.2, 0.2, E * A), rng.uniform(2.0, 6.0, E * A)], axis=1) for _ in range((T0 + STEPS) // 20 + 1)])
sim = orc.SimOracle(E, A); sim.set_map_dt(dt, res, origin)
sim.set_noise(np.random.default_rng(12345).normal(0., .01, size=(T0 + STEPS + 2, B)))
sim.reset(poses)
sim.rollout(sets, T0, 20, poses, True, 8)
theta_dis, fov = 2000, 4.7
sines = np.sin(np.linspace(0, 2 * np.pi, theta_dis)); cosines = np.cos(np.linspace(0, 2 * np.pi, theta_dis))
inc = theta_dis * (fov / (B - 1)) / (2 * np.pi)
H, W = dt.shape
N = E * A


def lookup(x, y):
    xt = x - origin[0]; yt = y - origin[1]
    oob = (xt < 0) | (xt >= W * res) | (yt < 0) | (yt >= H * res)
    cc = (xt / res).astype(np.int64); rr = (yt / res).astype(np.int64)
    cc[oob] = -1; rr[oob] = -1
    return dt[rr, cc]


def ray_counts(st):
    ti = theta_dis * (st[:, 4] - fov / 2.) / (2 * np.pi); ti = np.fmod(ti, theta_dis); ti[ti < 0] += theta_dis
    idx = np.empty((N, B), dtype=np.int64); cur = ti.copy()
    for b in range(B):
        idx[:, b] = cur.astype(np.int64)
        cur = cur + inc; cur[cur >= theta_dis] -= theta_dis
    idx = idx.reshape(-1); c = cosines[idx]; s = sines[idx]
    x = np.repeat(st[:, 0], B); y = np.repeat(st[:, 1], B)
    d = lookup(x, y); total = d.copy(); n = np.ones(x.shape, dtype=np.int32)
    active = np.nonzero((d > 1e-4) & (total <= 30.0))[0]
    while active.size:
        x[active] += d[active] * c[active]; y[active] += d[active] * s[active]
        dn = lookup(x[active], y[active]); d[active] = dn; total[active] += dn; n[active] += 1
        active = active[(dn > 1e-4) & (total[active] <= 30.0)]
    return n.reshape(N, B)


tasks = (B + 63) // 64
prev_max = None
print("# step  longest ray  rays>%d  of which: task listed / neighbour task listed / agent just re-seated / none of these" % LONG)
tot = np.zeros(5, dtype=np.int64)
for t in range(STEPS):
    k = (T0 + t) // 20
    sim.rollout(sets[k:k + 1], 1, 20, poses, True, 8)
    # (the scan of step t is traced from the pose AFTER the integration, i.e. the state the oracle holds now — unless the agent was
    # re-seated at the end of the step, which the NEXT step's scan sees; step_count == 0 marks those)
    n = ray_counts(sim.state.copy())
    reseated = np.repeat(sim.step_count == 0, 1)
    pad = np.zeros((N, tasks * 64), dtype=np.int32); pad[:, :B] = n
    tmax = pad.reshape(N, tasks, 64).max(axis=2)
    if prev_max is not None:
        ag, bm = np.nonzero(n > LONG)
        tk = bm // 64
        listed = prev_max[ag, tk] > THR
        nb = np.zeros_like(listed)
        for dlt in (-1, 1):
            q = np.clip(tk + dlt, 0, tasks - 1)
            nb |= prev_max[ag, q] > THR
        fresh = prev_reseated[ag]
        cat = np.array([len(ag), int(listed.sum()), int((~listed & nb).sum()), int((~listed & ~nb & fresh).sum()), int((~listed & ~nb & ~fresh).sum())])
        tot += cat
        print("%4d  %6d  %5d   %4d / %4d / %4d / %4d" % (t, n.max(), cat[0], cat[1], cat[2], cat[3], cat[4]))
    prev_max = tmax
    prev_reseated = reseated.copy()
print("# total rays > %d: %d; task listed %d, only a neighbour task listed %d, agent just re-seated %d, unpredicted %d" % (LONG, *tot))
