"""Timeline of ONE scan launch, wave by wave (experimental build: f110_exp_set scan_trace_hi / scan_trace_lo).
Every wave of k_scan_rays_agent records its begin / end on the 100 MHz clock, the CU it ran on and how many lock-step
samples it marched.  Answers: when do waves start, how long does each take against its samples, how full is the
chip over the launch, which waves end last.
    F110_LIB_VARIANT=experimental python tools/debug/scan_timeline.py [agents=4096] [task_order=-1 (default)] [steps=3] [tasks per wave=0 (default)]"""
import os, sys
os.environ.setdefault("F110_LIB_VARIANT", "experimental")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import f1tenth_gym_amd as amd
from _util import load_map_image
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
TORD = int(sys.argv[2]) if len(sys.argv) > 2 else -1
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
TPW = int(sys.argv[4]) if len(sys.argv) > 4 else 0
A, B = 2, 1080
E = N // A
exp = {} if TORD < 0 else {"task_order": TORD}
s = amd.BatchSim(num_envs=E, num_agents=A, exp=exp, scan_tasks_per_wave=TPW)
s.set_map_image(*load_map_image("example_map"))
s.set_noise_rng(12345, 0.01); s.noise_prepare(800)
poses0 = bench.start_poses_for(bench.shard_envs(E, 0), A).reshape(N, 3)
sets = bench.action_sets(40, N, seed=1000)
d_act = s.device_array((N, 2)); d_start = s.device_array((N, 3)); d_start.upload(poses0)
s.reset_device(d_start); s.set_auto_reseat(d_start, 0, None)
for t in range(500):
    if t % 20 == 0:
        d_act.upload(sets[t // 20])
    s.step_device(d_act)
s.sync()
n_waves = N * ((B + 63) // 64) + 65536
tr = s.device_array((n_waves, 8), dtype=np.uint64)

def set_trace(ptr):
    s.exp_set("scan_trace_hi", int(np.array(ptr >> 32, dtype=np.uint32).view(np.int32)))
    s.exp_set("scan_trace_lo", int(np.array(ptr & 0xffffffff, dtype=np.uint32).view(np.int32)))

for step in range(STEPS):
    tr.upload(np.zeros((n_waves, 8), dtype=np.uint64))
    set_trace(tr.ptr)
    s.step_device(d_act); s.sync()
    set_trace(0)
    for _ in range(7):
        s.step_device(d_act)
    s.sync()
    r = tr.download()
    live = r[:, 1] != 0
    idx = np.nonzero(live)[0]
    b = r[live, 0].astype(np.int64); e = r[live, 1].astype(np.int64)
    t0 = b.min()
    b = (b - t0) * 10.0; e = (e - t0) * 10.0   # ns
    hw = (r[live, 2] & 0xffffffff).astype(np.int64); xcc = (r[live, 2] >> 32).astype(np.int64) & 0xf
    samples = (r[live, 3] & 0xffffffff).astype(np.int64); lp = (r[live, 3] >> 32).astype(np.int64) != 0
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    dur = e - b
    ph = [(r[live, c].astype(np.int64) - t0) * 10.0 for c in (4, 5, 6, 7)]   # loop entered, header arrived, operands arrived, marched
    span = e.max()
    print("step %d: %d waves recorded (%d in the long pass), launch span %.1f us (first begin -> last end)" % (step, live.sum(), lp.sum(), span / 1e3))
    print("  distinct CUs seen: %d; waves per CU min/mean/max: %d / %.1f / %d" % (len(np.unique(cuid)), np.bincount(np.unique(cuid, return_inverse=True)[1]).min(),
          live.sum() / float(len(np.unique(cuid))), np.bincount(np.unique(cuid, return_inverse=True)[1]).max()))
    # concurrency over time
    edges = np.arange(0, span + 2000, 2000.0)
    line = []
    for t in edges[:-1]:
        running = ((b <= t + 1000) & (e > t + 1000)).sum()
        line.append("%d" % running)
    print("  waves running at t = 1, 3, 5, ... us: " + " ".join(line))
    started = np.histogram(b, bins=edges)[0]
    print("  waves started per 2 us:               " + " ".join("%d" % x for x in started))
    # duration against samples
    for lo_, hi_ in ((0, 10), (10, 20), (20, 35), (35, 65)):
        m = (b >= lo_ * 1e3) & (b < hi_ * 1e3) & ~lp & (samples > 0)
        if m.sum() > 10:
            sl, ic = np.polyfit(samples[m], dur[m], 1)
            print("  normal waves begun in [%2d, %2d) us: %6d   duration mean %.2f us (p50 %.2f, p99 %.2f)   samples mean %.1f   fit %.0f ns + %.0f ns per sample" % (
                lo_, hi_, m.sum(), dur[m].mean() / 1e3, np.median(dur[m]) / 1e3, np.percentile(dur[m], 99) / 1e3, samples[m].mean(), ic, sl))
    m = ~lp & (samples > 0) & (ph[2] > 0)
    if m.sum() > 10:
        seg = [ph[0][m] - b[m], ph[1][m] - ph[0][m], ph[2][m] - ph[1][m], ph[3][m] - ph[2][m], e[m] - ph[3][m]]
        names = ["begin -> task loop (kernel arguments, block remap)", "-> stamp + header arrived", "-> direction / noise arrived (march begins)", "-> marched", "-> end (store, iTTC, listing)"]
        print("  where a normal wave's time goes (mean / p50 / p90, ns):")
        for nm, sg in zip(names, seg):
            print("     %-52s %6.0f / %6.0f / %6.0f" % (nm, sg.mean(), np.median(sg), np.percentile(sg, 90)))
        print("     march per sample: %.0f ns;  gap between a slot's waves (launch span * slots / waves - duration): %.0f ns" % (
            seg[3].sum() / max(samples[m].sum(), 1), span * 256 * 4 * 8 / float(live.sum()) - dur.mean()))
    if lp.any():
        m = lp
        sl, ic = np.polyfit(samples[m], dur[m], 1)
        print("  long-pass waves: %d  begin %.1f..%.1f us  samples mean %.0f max %d  duration mean %.1f max %.1f us  fit %.0f ns + %.0f ns per sample" % (
            m.sum(), b[m].min() / 1e3, b[m].max() / 1e3, samples[m].mean(), samples[m].max(), dur[m].mean() / 1e3, dur[m].max() / 1e3, ic, sl))
    last = np.argsort(e)[-12:][::-1]
    print("  the waves that end last:")
    for i in last:
        print("     end %.1f us  begin %.1f  duration %.1f  samples %d  %s  launch-order wave %d  cu %d" % (e[i] / 1e3, b[i] / 1e3, dur[i] / 1e3, samples[i], "long pass" if lp[i] else "normal", idx[i], cuid[i]))
    for q in (50, 90, 99, 99.9):
        print("  %.1f %% of the waves have ended by %.1f us" % (q, np.percentile(e, q) / 1e3), end=";")
    print()
s.close()
