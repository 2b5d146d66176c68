"""Rates of the drop-in paths (what an RL loop calls), next to the kernel path:
  F110Env(num_agents=2).step()  — 1 env, host actions, the reference's obs dict (BASELINE configs[0] on the GPU)
  F110VecEnv(E, device_logic=True) — default episode fields / lean, actions copied / written in place
  BatchSim.step_device            — kernels only, nothing crosses PCIe
usage: dropin_rate.py [E,E,...]"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import f1tenth_gym_amd as amd
from f1tenth_gym_amd import build
from _util import MAPS, bench_start_poses

MAP = dict(map=os.path.join(MAPS, "example_map"), map_ext=".png")
print("# csrc %s  tools/debug/dropin_rate.py" % build.src_hash())


def timed(fn, n, warm=20):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n


# ---- the drop-in itself: 1 env x 2 agents through F110Env.step (host actions, obs dict of lists)
env = amd.F110Env(num_agents=2, **MAP)
poses = bench_start_poses(1, 2).reshape(2, 3)
env.reset(poses)
act = np.array([[0.05, 3.0], [-0.05, 2.5]])
dt = timed(lambda: env.step(act), 2000, 100)
c, enq, wait = env.sim.batch.step_host_stats()
print("F110Env(num_agents=2).step              %8.1f us/step  %9.0f env-steps/s  %9.0f agent-steps/s   [host: enqueue %.1f us, wait %.1f us, Python around the call %.1f us]"
      % (dt * 1e6, 1 / dt, 2 / dt, enq, wait, dt * 1e6 - enq - wait))
# the same env through the raw one-call-per-step path (no obs dict, no lap logic): what the ABI itself costs
b = env.sim.batch
hb = b.host_block(("scans", "state", "collisions"))
hb.actions[...] = act
dt2o = timed(lambda: b.step_host(hb, poll=False), 2000, 100)
c, enq, wait = b.step_host_stats()
print("  BatchSim.step_host (same env, scans, hipStreamSynchronize)   %8.1f us/step   [host: enqueue %.1f us, wait %.1f us]" % (dt2o * 1e6, enq, wait))
dt2 = timed(lambda: b.step_host(hb), 2000, 100)
c, enq, wait = b.step_host_stats()
print("  BatchSim.step_host (same env, scans+state+collisions)  %8.1f us/step   [host: enqueue %.1f us, wait %.1f us]" % (dt2 * 1e6, enq, wait))
dt2s = timed(lambda: b.step_host(hb, spin=True), 2000, 100)
c, enq, wait = b.step_host_stats()
print("  BatchSim.step_host (same env, scans, completion word polled) %8.1f us/step   [host: enqueue %.1f us, wait %.1f us]" % (dt2s * 1e6, enq, wait))
hb2 = b.host_block(("state", "collisions"))
hb2.actions[...] = act
dt3 = timed(lambda: b.step_host(hb2), 2000, 100)
c, enq, wait = b.step_host_stats()
print("  BatchSim.step_host (same env, no scans)                %8.1f us/step   [host: enqueue %.1f us, wait %.1f us]" % (dt3 * 1e6, enq, wait))

sizes = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2048, 32768]
for E in sizes:
    N = E * 2
    poses = bench_start_poses(E, 2).reshape(E, 2, 3)
    rng = np.random.default_rng(0)
    act = np.stack([rng.uniform(-0.2, 0.2, (E, 2)), rng.uniform(2, 6, (E, 2))], axis=2)
    n = 300 if E <= 4096 else 100
    # kernel path
    env = amd.F110VecEnv(E, auto_reset=True, device_logic=True, obs_fields=(), **MAP)
    env.reset(poses)
    b = env.sim.batch
    d_act = b.device_array((N, 2)); d_act.upload(act.reshape(N, 2))
    def dev_steps(k=50):
        for _ in range(k):
            b.episode_step_device(d_act)
            b.episode_reset_done_device()
        b.sync()
    dev_steps(); t0 = time.perf_counter(); dev_steps(n); dk = (time.perf_counter() - t0) / n
    print("E=%6d  device-resident loop (episode_step_device + reset_done, no sync)   %.4f ms/step  %6.2f M agent-steps/s" % (E, dk * 1e3, N / dk / 1e6))
    b.close()
    for label, kw, inplace in (("VecEnv default episode fields, step(actions)", {}, False),
                               ("VecEnv default episode fields, in-place actions", {}, True),
                               ("VecEnv episode_fields=(), step(actions)", {"episode_fields": ()}, False),
                               ("VecEnv episode_fields=(), in-place actions", {"episode_fields": ()}, True),
                               ("VecEnv episode_fields=(), in-place, staged H2D", {"episode_fields": (), "mapped_actions": False}, True),
                               ("VecEnv episode_fields=(), in-place, spin wait", {"episode_fields": (), "spin_wait": True}, True),
                               ("VecEnv default episode fields, step(actions), spin", {"spin_wait": True}, False),
                               ("VecEnv episode_fields=(), in-place, hipStreamSynchronize", {"episode_fields": (), "poll_wait": False}, True),
                               ("VecEnv obs poses+collisions, default episode", {"obs_fields": ("poses_x", "poses_y", "poses_theta", "collisions")}, False)):
        kw = dict(kw); kw.setdefault("obs_fields", ())
        env = amd.F110VecEnv(E, auto_reset=True, device_logic=True, **kw, **MAP)
        env.reset(poses)
        if inplace:
            env.action_buffer[...] = act
            dt = timed(lambda: env.step(None), n)
        else:
            dt = timed(lambda: env.step(act), n)
        c, enq, wait = env.sim.batch.step_host_stats()
        print("E=%6d  %-52s %.4f ms/step  %6.2f M agent-steps/s   [host: enqueue %.1f us, wait %.1f us]" % (E, label, dt * 1e3, N / dt / 1e6, enq, wait))
        env.sim.batch.close()
