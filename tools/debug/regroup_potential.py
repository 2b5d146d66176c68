"""How much would the scan gain if an agent's beams were grouped into 64-ray tasks by how long their rays were
in the PREVIOUS step, instead of by beam number?  A task's march runs until its longest ray ends (lock-step lanes),
so its cost is max over the 64 lanes of the lookups; neighbouring beams are correlated but not equal.
CPU only: the oracle rolls the bench workload to its steady regime, then the per-beam lookup counts of two
consecutive steps are taken with the oracle's trace_ray.
    python tools/debug/regroup_potential.py [envs=256] [steps=450]"""
import ctypes as C, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import orc
from _util import oracle_map_dt
import bench

E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 450
A, B = 2, 1080
orc.build()
L = orc.lib()
L.orc_get_scan_counts.argtypes = [C.POINTER(orc.ScanCfg), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
dt, res, origin = oracle_map_dt("example_map")
ref = orc.SimOracle(E, A); ref.set_map_dt(dt, res, origin)
so = orc.ScanOracle(B, 4.7); so.set_map_dt(dt, res, origin)
poses = bench.start_poses_for(bench.shard_envs(E, 0), A)
ref.reset(poses)
sets = bench.action_sets((T + 21) // 20 + 1, E * A, seed=1000)
t0 = time.time()
snap = []
for t in range(T + 1):
    ref.step(sets[t // 20], os.cpu_count() or 1)
    if t >= T - 1:
        snap.append(ref.agent_poses.copy())
    mask = (ref.collisions.reshape(E, A)[:, 0] != 0).astype(np.uint8)
    if mask.any() and t < T - 1:      # (no re-seat between the two snapshots)
        ref.reset(poses, mask)
print("rolled %d steps of %d agents in %.1f s" % (T + 1, E * A, time.time() - t0))

def counts_of(poses3):
    out = np.empty((poses3.shape[0], B), dtype=np.int32)
    for i, p in enumerate(np.ascontiguousarray(poses3)):
        L.orc_get_scan_counts(C.byref(so.cfg), p.ctypes.data_as(C.POINTER(C.c_double)), out[i].ctypes.data_as(C.POINTER(C.c_int32)))
    return out
prev, cur = counts_of(snap[0]), counts_of(snap[1])
print("mean lookups per ray: %.3f (this step), %.3f (previous)" % (cur.mean(), prev.mean()))
pad = (-B) % 64
def task_cost(c, order=None):
    """sum over 64-ray tasks of the longest ray (the first lookup is shared by the agent, the loop counts the rest)"""
    if order is not None:
        c = np.take_along_axis(c, order, axis=1)
    c = np.concatenate([c, np.zeros((c.shape[0], pad), dtype=c.dtype)], axis=1).reshape(c.shape[0], -1, 64)
    return c.max(axis=2).sum() / float(c.shape[0] * c.shape[1])
base = task_cost(cur)
print("gathers per task, beams in beam order (what the kernel does):       %.2f" % base)
print("  ... sorted by THIS step's lengths (unattainable bound):            %.2f" % task_cost(cur, np.argsort(cur, axis=1, kind="stable")))
print("  ... sorted by the PREVIOUS step's lengths:                        %.2f" % task_cost(cur, np.argsort(prev, axis=1, kind="stable")))
for bins in (2, 4, 8):
    # coarse: beams keep their order inside `bins` classes of the previous step's length (quantiles)
    q = np.quantile(prev, np.linspace(0, 1, bins + 1)[1:-1], axis=1).T            # [agents][bins-1]
    cls = (prev[:, :, None] > q[:, None, :]).sum(axis=2)
    print("  ... %d classes of the previous step's length, beam order inside:   %.2f" % (bins, task_cost(cur, np.argsort(cls, axis=1, kind="stable"))))
blk = prev.reshape(prev.shape[0], -1, 8).max(axis=2) if B % 8 == 0 else None
if blk is not None:
    # groups of 8 consecutive beams (one 64-byte piece of the row: stores stay coalesced) sorted by their previous maximum
    o8 = np.argsort(blk, axis=1, kind="stable")
    order = (o8[:, :, None] * 8 + np.arange(8)[None, None, :]).reshape(prev.shape[0], -1)
    print("  ... groups of 8 consecutive beams sorted by their previous maximum:  %.2f" % task_cost(cur, order))
print("(the march is 9.6 of the kernel's 13.2 vector-memory instructions per task: a task cost of X means %.0f %% of today's instructions at X = ...)" % 100.0)
for x in (8.0, 7.0, 6.5):
    print("   X = %.1f -> %.0f %%" % (x, 100.0 * (13.2 - (base - 1 - (x - 1))) / 13.2))
