"""single env (two parked cars), raw f110_step_host loop: what does host time BETWEEN the calls cost?  (a) back to back, (b) a busy wait of
12 us between calls, (c) a NumPy copy of the scans out of the page-locked block between calls, (d) both"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import f1tenth_gym_amd as amd
from _util import load_map_image, bench_start_poses
img, res, origin = load_map_image("example_map")
b = amd.BatchSim(num_envs=1, num_agents=2)
b.set_map_image(img, res, origin); b.set_noise_rng(12345, 0.01)
b.reset(bench_start_poses(1, 2))
hb = b.host_block(("scans", "state", "collisions"))
hb.actions[...] = 0.0     # parked cars: the same rays every step, so the variants are comparable


def busy(us):
    t = time.perf_counter() + us * 1e-6
    while time.perf_counter() < t:
        pass


def loop(n, gap_us=0.0, copy=False, poll=True, block=None):
    blk = hb if block is None else block
    b.step_host_stats()
    t0 = time.perf_counter()
    for _ in range(n):
        b.step_host(blk, poll=poll)
        if copy:
            x = hb.views["scans"].copy()
        if gap_us:
            busy(gap_us)
    dt = (time.perf_counter() - t0) / n
    c, enq, wait = b.step_host_stats()
    return dt * 1e6, enq, wait


hb_noscan = b.host_block(("state", "collisions"))
hb_noscan.actions[...] = 0.0
for label, kw in (("back to back", {}), ("12 us busy wait between calls", {"gap_us": 12.0}), ("30 us busy wait", {"gap_us": 30.0}),
                  ("scans copied out between calls", {"copy": True}), ("copy + 12 us", {"copy": True, "gap_us": 12.0}), ("back to back again", {}),
                  ("back to back, hipStreamSynchronize", {"poll": False}), ("copy + 12 us, hipStreamSynchronize", {"poll": False, "copy": True, "gap_us": 12.0}),
                  ("no scans in the block", {"block": hb_noscan}), ("no scans, hipStreamSynchronize", {"block": hb_noscan, "poll": False})):
    loop(300, **kw)
    dt, enq, wait = loop(2000, **kw)
    print("%-34s %6.1f us per iteration   [in the call: enqueue %.1f us, wait %.1f us]" % (label, dt, enq, wait))

# per-call distribution (is the mean a mix of discrete modes?)
for label, kw in (("back to back", {}), ("12 us busy wait between calls", {"gap_us": 12.0}), ("30 us busy wait", {"gap_us": 30.0})):
    ts = []
    for _ in range(3000):
        t0 = time.perf_counter()
        b.step_host(hb)
        ts.append((time.perf_counter() - t0) * 1e6)
        if kw.get("gap_us"):
            busy(kw["gap_us"])
    ts = np.array(ts[500:])
    hist, edges = np.histogram(ts, bins=[0, 30, 35, 40, 45, 50, 60, 70, 80, 90, 100, 120, 1e9])
    print("%-32s call us: p10 %.1f p50 %.1f p90 %.1f   histogram %s" % (label, np.percentile(ts, 10), np.percentile(ts, 50), np.percentile(ts, 90),
          " ".join("%s:%d" % (("<%d" % edges[i + 1]) if edges[i + 1] < 1e8 else ">120", h) for i, h in enumerate(hist) if h)))
    # autocorrelation at lag 1: do slow and fast calls alternate?
    z = ts - ts.mean()
    print("    lag-1 autocorrelation %.2f   first 24 calls: %s" % (float((z[1:] * z[:-1]).sum() / (z * z).sum()), " ".join("%.0f" % v for v in ts[:24])))
