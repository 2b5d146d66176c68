"""Survivor compaction in k_scan_rays_agent, priced on the CPU before it is built.

The scan marches 64 consecutive beams of one agent per wave pass in lock step: a pass costs the lookups of its LONGEST
ray (lane utilisation 0.62 at the headline workload).  This script takes the rays of the bench workload's steady regime
(oracle rollout, then a vectorised NumPy restatement of trace_ray that records the table cell of every sample), and counts,
per scheme, the wave-level gather instructions the march issues and the 128-byte lines each of them touches:

  today            every 64-beam task runs until its longest ray ends
  two-phase(K, D)  every task of a domain D (the 3 tasks of a wave / the 12 of a workgroup / the 17 of an agent) marches at
                   most K samples; the rays still alive are packed densely (beam order) into new 64-lane groups through
                   LDS, which march to the end
  three-phase      the same with a second packing at K2

Cost models per wave-level gather touching L distinct lines (tools/debug/ta_bench.hip, profiles/r0*_ta_bench.txt: a u64
gather costs 19-20 CU cycles up to 8-16 lines, ~33 at 32, 64-96 at 64):  flat = 1 per gather;  ta = max(19.5, L) cycles;
ta14 = max(19.5, 1.4 L).   Packing itself is not free: `pack` = the LDS round trip + bookkeeping per packed task, given in
gather equivalents (default 2 per task that takes part, i.e. ~40 cycles).

usage: python tools/debug/compaction_sim.py [envs=256] [steps=320]     (about a minute; output kept in profiles/r05_compaction_sim.txt)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
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
TPA = (B + 63) // 64                       # tasks per agent: 17 (the last one holds 56 beams)
ti = theta_dis * (st[:, 4] - fov / 2.) / (2 * np.pi); ti = np.fmod(ti, theta_dis); ti[ti < 0] += theta_dis
idx = np.empty((N, B), dtype=np.int64)
cur = ti.copy()
for b in range(B):
    idx[:, b] = cur.astype(np.int64)
    cur = cur + inc; cur[cur >= theta_dis] -= theta_dis
idx = idx.reshape(-1); c = cosines[idx]; s = sines[idx]
H, W = dt.shape
PADB = int(np.ceil(30.0 / res)) + 66       # the PADDED layout's border (f110_hip.hip build_padded)
PW = W + 2 * PADB


def lookup(x, y):
    xt = x - origin[0]; yt = y - origin[1]
    oob = (xt < 0) | (xt >= W * res) | (yt < 0) | (yt >= H * res)
    cc = (xt / res).astype(np.int64); rr = (yt / res).astype(np.int64)
    line = ((rr + PADB) * PW + (cc + PADB)) >> 4           # 16 float64 cells per 128-byte line of the padded table
    cc[oob] = -1; rr[oob] = -1
    return dt[rr, cc], line


xs = np.repeat(st[:, 0], B); ys = np.repeat(st[:, 1], B)
agent = np.repeat(np.arange(N), B); beam = np.tile(np.arange(B), N)
task0 = agent * TPA + beam // 64           # today's grouping
d, _ = lookup(xs, ys)                      # the first sample belongs to k_integrate (RayHdr.d0): not a gather of the scan kernel
total = d.copy()
active = (d > 1e-4) & (total <= 30.0)
x = xs.copy(); y = ys.copy()
# pass 1: gathers per ray (n) and, per iteration, the active rays and the lines they read
iters = []          # list of (ray ids, lines)
n = np.zeros(x.shape, dtype=np.int64)
while active.any():
    ia = np.nonzero(active)[0]
    x[ia] += d[ia] * c[ia]; y[ia] += d[ia] * s[ia]
    dn, line = lookup(x[ia], y[ia]); d[ia] = dn; total[ia] += dn; n[ia] += 1
    iters.append((ia.astype(np.int32), line))
    active[ia] = (dn > 1e-4) & (total[ia] <= 30.0)
print("# rays %d, march gathers per ray: mean %.2f (+1 first sample in k_integrate), max %d; iterations %d" % (x.size, n.mean(), n.max(), len(iters)))
nt = n.reshape(N, B)
tmax = np.array([nt[:, k:k + 64].max(axis=1) for k in range(0, B, 64)]).T.reshape(-1)
print("# tasks %d: mean of the task maximum %.2f, p50 %d p90 %d p99 %d max %d; lane utilisation today %.3f"
      % (tmax.size, tmax.mean(), np.percentile(tmax, 50), np.percentile(tmax, 90), np.percentile(tmax, 99), tmax.max(), n.sum() / (64.0 * tmax.sum())))


def phase2_groups(K, tasks_per_domain):
    """rays with more than K gathers, packed in beam order into groups of 64 inside their domain; -> group id per ray (-1: done by K)"""
    surv = np.nonzero(n > K)[0]
    if tasks_per_domain >= TPA:
        dom = agent[surv]
    else:
        # a wave walks `tpw` CONSECUTIVE tasks of the global task order (tasks_per_wave = 3), a workgroup 4 waves' worth
        dom = task0[surv] // tasks_per_domain
    # survivors are already sorted by (agent, beam) = by domain, beam order
    first = np.r_[True, dom[1:] != dom[:-1]]
    start = np.maximum.accumulate(np.where(first, np.arange(len(surv)), 0))
    rank = np.arange(len(surv)) - start
    dom_index = np.cumsum(first) - 1
    g = np.full(x.shape, -1, dtype=np.int64)
    g[surv] = dom_index * 64 + rank // 64          # (a domain holds at most 17 tasks: < 64 groups)
    return g, len(np.unique(dom)), len(surv)


def price(group_of_iter, label, pack_tasks=0, pack_cost=2.0):
    """group_of_iter(k) -> array ray -> group id at march iteration k (1-based)"""
    gathers = 0; lines_sum = 0; ta = 0.0; ta14 = 0.0
    for k, (ia, line) in enumerate(iters, start=1):
        g = group_of_iter(k)[ia]
        key = g * (1 << 34) + line
        uk = np.unique(key)
        per_group = np.bincount(np.unique(uk >> 34, return_inverse=True)[1])     # distinct lines per issued gather
        gathers += per_group.size; lines_sum += per_group.sum()
        ta += np.maximum(19.5, per_group).sum(); ta14 += np.maximum(19.5, 1.4 * per_group).sum()
    return dict(label=label, gathers=gathers, lines=lines_sum, ta=ta + 19.5 * pack_cost * pack_tasks, ta14=ta14 + 19.5 * pack_cost * pack_tasks,
                flat=gathers + pack_cost * pack_tasks)


rows = []
base = price(lambda k: task0, "today (64 consecutive beams, run to the longest ray)")
rows.append(base)
for dom_name, tpd in (("wave: 3 tasks", 3), ("workgroup: 12 tasks", 12), ("agent: 17 tasks", TPA)):
    for K in (2, 4, 6, 8, 12, 16):
        g2, n_dom, n_surv = phase2_groups(K, tpd)
        # packing is paid by every task that still has a survivor at K (its lanes write their state to LDS) — and read back by the packed groups
        tasks_with_surv = len(np.unique(task0[n > K]))
        rows.append(price(lambda k, g2=g2, K=K: task0 if k <= K else g2, "two-phase K=%-2d %-20s survivors %4.1f %% of rays" % (K, dom_name, 100.0 * n_surv / x.size),
                          pack_tasks=tasks_with_surv))
def multi_phase(levels, tpd, name):
    """pack at every K of `levels`: groups after level i come from phase2_groups(levels[i]); the tasks that pay for a packing
    are the groups of the previous phase that still hold a survivor"""
    maps = [task0] + [phase2_groups(K, tpd)[0] for K in levels]
    pack = sum(len(np.unique(maps[i][n > K])) for i, K in enumerate(levels))
    bounds = np.array(levels)

    def group_of(k):
        return maps[int(np.searchsorted(bounds, k, side="left"))]      # k <= levels[0]: today's tasks; levels[i-1] < k <= levels[i]: maps[i]
    return price(group_of, "%d-phase K=%s %s" % (len(levels) + 1, ",".join(str(v) for v in levels), name), pack_tasks=pack)


for dom_name, tpd in (("workgroup: 12 tasks", 12), ("agent: 17 tasks", TPA)):
    for levels in ((4, 12), (6, 16), (6, 24), (8, 32), (4, 8, 16, 32, 64), (8, 16, 32, 64), (6, 12, 24, 48, 96), (3, 6, 12, 24, 48, 96), (16, 32, 64), (12, 48)):
        rows.append(multi_phase(levels, tpd, dom_name))
perfect_gathers = sum((len(ia) + 63) // 64 for ia, _ in iters)
print("# perfect packing at every sample (bound, no cost): %d gathers = %.3f of today" % (perfect_gathers, perfect_gathers / base["gathers"]))
print("%-78s %9s %7s %7s %7s %7s %7s" % ("scheme", "gathers", "vs", "lines/g", "flat+pk", "ta+pk", "ta14+pk"))
for r in rows:
    print("%-78s %9d %7.3f %7.2f %7.3f %7.3f %7.3f" % (r["label"], r["gathers"], r["gathers"] / base["gathers"], r["lines"] / r["gathers"],
                                                      r["flat"] / base["flat"], r["ta"] / base["ta"], r["ta14"] / base["ta14"]))


# ---------------------------------------------------------------------------------------------------------------------
# streaming refill: a wave owns a QUEUE of rays (q_tasks consecutive tasks) and re-fills its free lanes from the queue
# whenever at least `refill` of them are free — what "perfect packing" means for one wave.  Rays of a queue are taken in
# beam order, or longest-first by their TRUE length (the best any "last step's lengths" heuristic could do).
seq_off = np.zeros(x.size + 1, dtype=np.int64); np.cumsum(n, out=seq_off[1:])
seq = np.empty(int(seq_off[-1]), dtype=np.int64)
fill = seq_off[:-1].copy()
for ia, line in iters:
    seq[fill[ia]] = line; fill[ia] += 1


def stream(q_tasks, refill, order="beam"):
    n_q = (N * TPA + q_tasks - 1) // q_tasks
    queues = []
    for q in range(n_q):
        rays = np.nonzero((task0 >= q * q_tasks) & (task0 < (q + 1) * q_tasks))[0] if q_tasks != TPA else np.arange(q * B, (q + 1) * B)
        rays = rays[n[rays] > 0]
        if order == "longest":
            rays = rays[np.argsort(-n[rays], kind="stable")]
        queues.append(rays)
    head = np.zeros(n_q, dtype=np.int64)
    lane_ray = np.full((n_q, 64), -1, dtype=np.int64); lane_k = np.zeros((n_q, 64), dtype=np.int64)
    gathers = 0; lines_sum = 0; ta = 0.0; events = 0
    while True:
        free = lane_ray < 0
        nfree = free.sum(axis=1)
        for q in np.nonzero(((nfree >= refill) | (nfree == 64)) & (head < np.array([len(v) for v in queues])))[0]:
            take = min(int(nfree[q]), len(queues[q]) - int(head[q]))
            slots = np.nonzero(free[q])[0][:take]
            lane_ray[q, slots] = queues[q][head[q]:head[q] + take]; lane_k[q, slots] = 0
            head[q] += take; events += 1
        act = lane_ray >= 0
        if not act.any():
            break
        qi, li = np.nonzero(act)
        r = lane_ray[qi, li]
        line = seq[seq_off[r] + lane_k[qi, li]]
        uk = np.unique(qi.astype(np.int64) * (1 << 34) + line)
        per = np.bincount(np.unique(uk >> 34, return_inverse=True)[1])
        gathers += per.size; lines_sum += per.sum(); ta += np.maximum(19.5, per).sum()
        lane_k[qi, li] += 1
        done = lane_k[qi, li] >= n[r]
        lane_ray[qi[done], li[done]] = -1
    return gathers, lines_sum, ta, events


print("\n# streaming refill (one wave per queue; refill when >= R lanes are free); events = refills (each: finished ranges to LDS, new rays from LDS)")
print("%-64s %9s %7s %7s %7s %9s" % ("scheme", "gathers", "vs", "lines/g", "ta", "events/task"))
for q_tasks, qname in ((3, "queue = 3 tasks (a wave today)"), (TPA, "queue = one agent (17 tasks)"), (2 * TPA, "queue = two agents")):
    for order in ("beam", "longest"):
        for R in (1, 8, 16, 32):
            g, ls, ta, ev = stream(q_tasks, R, order)
            print("%-64s %9d %7.3f %7.2f %7.3f %9.2f" % ("%s, %s order, R=%d" % (qname, order, R), g, g / base["gathers"], ls / g, ta / base["ta"], ev / (N * TPA)))
