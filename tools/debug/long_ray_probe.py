"""What pins the 4096-agent scan: the longest ray's dependent chain, and at what price per sample?
Fixed poses (a steady-state snapshot of the bench workload, speed command 0 so nothing moves), the step's own scan
kernel timed with its event pair (f110_profile_kernels).  Variants: all agents; the agents with a long ray replaced
by a short-ray agent; ONLY the agents with a long ray (an otherwise idle chip).
    python tools/debug/long_ray_probe.py [envs=2048]"""
import ctypes as C, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import f1tenth_gym_amd as amd
from oracle import orc
from _util import load_map_image, oracle_map_dt
import bench

E = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
A, B = 2, 1080
N = E * A
img = load_map_image("example_map")

def make(n_envs, exp=None):
    s = amd.BatchSim(num_envs=n_envs, num_agents=A, exp=exp)
    s.set_map_image(*img)
    s.set_noise_rng(12345, 0.01); s.noise_prepare(800)
    return s

# 1. steady-state snapshot
s = make(E)
poses0 = bench.start_poses_for(bench.shard_envs(E, 0), A).reshape(N, 3)
sets = bench.action_sets(30, N, seed=1000)
d_act = s.device_array((N, 2)); d_start = s.device_array((N, 3)); d_start.upload(poses0)
s.reset_device(d_start); s.set_auto_reseat(d_start, 0, None)
for t in range(500):
    if t % 20 == 0:
        d_act.upload(sets[t // 20])
    s.step_device(d_act)
s.sync()
g = s.get("poses_x", "poses_y", "poses_theta")
P = np.stack([g["poses_x"].reshape(-1), g["poses_y"].reshape(-1), g["poses_theta"].reshape(-1)], axis=1)
s.close()

# 2. per-ray lookup counts at P (oracle as the analysis aid)
orc.build(); L = orc.lib()
L.orc_get_scan_counts.argtypes = [C.POINTER(orc.ScanCfg), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
dt, res, origin = oracle_map_dt("example_map")
so = orc.ScanOracle(B, 4.7); so.set_map_dt(dt, res, origin)
cnt = np.empty((N, B), dtype=np.int32)
for i, p in enumerate(np.ascontiguousarray(P)):
    L.orc_get_scan_counts(C.byref(so.cfg), p.ctypes.data_as(C.POINTER(C.c_double)), cnt[i].ctypes.data_as(C.POINTER(C.c_int32)))
amax = cnt.max(axis=1)
print("agents %d  mean lookups %.2f  longest ray %d  agents with a ray > 60/100/150/200: %d %d %d %d" % (
    N, cnt.mean(), amax.max(), (amax > 60).sum(), (amax > 100).sum(), (amax > 150).sum(), (amax > 200).sum()))
short_pose = P[np.argsort(amax)[N // 2]]

def timed(poses, label, exp=None, steps=200):
    n = poses.shape[0]
    n_envs = n // A
    s = make(n_envs, exp)
    d_p = s.device_array((n, 3)); d_p.upload(np.ascontiguousarray(poses))
    d_a = s.device_array((n, 2)); d_a.upload(np.zeros((n, 2)))
    s.reset_device(d_p)
    for _ in range(20):
        s.step_device(d_a)
    s.sync()
    s.profile_kernels(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        s.step_device(d_a)
    s.sync(); wall = (time.perf_counter() - t0) / steps
    nl, sm, dm, fm = s.profile_read()
    pr = {"launches": max(nl, 1), "scan_ms": sm, "dyn_ms": dm, "finalize_ms": fm}
    s.profile_kernels(False)
    g = s.get("poses_x")
    moved = float(np.abs(g["poses_x"].reshape(-1) - poses[:, 0]).max())
    print("%-46s agents %5d  scan %.1f us  integrate %.1f  finalize %.1f  (step wall %.1f us, moved %.1e)" % (
        label, n, 1e3 * pr["scan_ms"] / pr["launches"], 1e3 * pr["dyn_ms"] / pr["launches"], 1e3 * pr["finalize_ms"] / pr["launches"], wall * 1e6, moved))
    s.close()
    return 1e3 * pr["scan_ms"] / pr["launches"]

t_all = timed(P, "all agents (longest ray %d)" % amax.max())
print("   -> %.0f ns per sample of the longest ray if it alone pinned the kernel" % (1e3 * t_all / amax.max()))
for thr in (200, 150, 100, 60, 30):
    Q = P.copy(); Q[amax > thr] = short_pose
    timed(Q, "agents with a ray > %d replaced (%d)" % (thr, (amax > thr).sum()))
for thr in (150, 100):
    idx = np.nonzero(amax > thr)[0]
    k = (len(idx) // 2) * 2
    if k >= 2:
        sub = P[idx[:k]]
        t = timed(sub, "ONLY the agents with a ray > %d (longest %d)" % (thr, amax[idx[:k]].max()))
        print("   -> %.0f ns per sample on the otherwise idle chip" % (1e3 * t / amax[idx[:k]].max()))
# the single longest agent with a partner
i = int(np.argmax(amax))
t = timed(np.stack([P[i], short_pose]), "the agent with the longest ray + one other")
print("   -> %.0f ns per sample" % (1e3 * t / amax[i]))
