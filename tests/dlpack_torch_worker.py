"""test_gpu_round5.py::test_dlpack_hand_off_to_torch, in a process of its own (torch first, then the simulator's library)."""
import os
import sys

try:
    import torch
except Exception as ex:  # noqa: BLE001
    print("SKIP torch is not importable: %s" % ex)
    sys.exit(0)
if not torch.cuda.is_available():
    print("SKIP this torch build sees no GPU")
    sys.exit(0)
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import f1tenth_gym_amd as amd  # noqa: E402
from _util import bench_start_poses, load_map_image  # noqa: E402

E, A = 16, 2
N = E * A
s = amd.BatchSim(num_envs=E, num_agents=A)
s.set_map_image(*load_map_image("example_map"))
s.reset(bench_start_poses(E, A))
views = s.device_views()
assert views["scans"].__dlpack_device__() == (10, 0)
act = s.device_array((N, 2)); act.upload(np.zeros((N, 2)))
s.step_device(act)
scans_t = torch.from_dlpack(views["scans"])
act_t = torch.from_dlpack(act)
assert scans_t.data_ptr() == views["scans"].ptr and act_t.data_ptr() == act.ptr and scans_t.dtype == torch.float64
assert tuple(scans_t.shape) == (N, 1080) and scans_t.is_contiguous() and scans_t.device.type == "cuda"
assert np.array_equal(scans_t.cpu().numpy(), s.get("scans")["scans"])
act_t[:, 0] = 0.1
act_t[:, 1] = 3.0
torch.cuda.synchronize()
ref = amd.BatchSim(num_envs=E, num_agents=A)
ref.set_map_image(*load_map_image("example_map")); ref.reset(bench_start_poses(E, A))
ref.step(np.zeros((N, 2)))
for _ in range(5):
    s.step_device(act)
    ref.step(np.tile([0.1, 3.0], (N, 1)))
assert np.array_equal(s.get("state")["state"], ref.get("state")["state"])
assert np.array_equal(scans_t.cpu().numpy(), ref.get("scans")["scans"])     # the same buffer, the new step's values
# the int32 / uint8 views too
assert torch.from_dlpack(views["in_collision"]).dtype == torch.int32
del scans_t, act_t
torch.cuda.synchronize()
s.close(); ref.close()

# ---- the documented RL loop (examples/rl_loop_device.py --torch): torch work on an ExternalStream around the
# handle's main stream, fenced (f110_stream_fence), with the step free to go out as two env blocks (step_groups 0 / 2) — against
# the same loop on a one-block handle (step_groups 1), where stream order alone is enough.  8192 agents: a size at which the
# automatic mode splits back-to-back steps.
def rl_loop(groups, steps=40):
    E2, A2 = 4096, 2
    N2 = E2 * A2
    b = amd.BatchSim(num_envs=E2, num_agents=A2, step_groups=groups)
    b.set_map_image(*load_map_image("example_map")); b.set_noise_rng(12345, 0.01)
    b.episode_init(0); b.episode_reset(bench_start_poses(E2, A2))
    actions = b.device_array((N2, 2)); actions.upload(np.zeros((N2, 2)))
    v = b.device_views()
    stream = torch.cuda.ExternalStream(v["stream"], device=torch.device("cuda", b.device_id))
    sc, ac = torch.from_dlpack(v["scans"]), torch.from_dlpack(actions)
    b.episode_step_device(actions)
    blocks = set()
    for _ in range(steps):
        b.fence()
        with torch.cuda.stream(stream):
            ahead = sc[:, 500:580].min(dim=1).values                       # reads the step's scans ...
            side = sc[:, 700:900].mean(dim=1) - sc[:, 180:380].mean(dim=1)
            ac[:, 0] = torch.clamp(0.05 * side, -0.4, 0.4)                 # ... writes the next step's actions
            ac[:, 1] = torch.clamp(ahead, 1.0, 6.0)
        b.episode_step_device(actions); b.episode_reset_done_device()
        b.episode_step_device(actions); b.episode_reset_done_device()      # two steps per policy call: the second may split
        blocks.add(b.step_groups()[2])
    out = b.get("state", "scans", "collisions")
    out["laps"] = b.episode_get()["lap_counts"]
    del sc, ac
    torch.cuda.synchronize()
    b.close()
    return out, blocks

base, _ = rl_loop(1)
for gmode in (0, 2):
    got, blocks = rl_loop(gmode)
    for key in base:
        assert np.array_equal(base[key], got[key]), (gmode, key)
    print("RL LOOP groups=%d blocks seen %s OK" % (gmode, sorted(blocks)))
print("DLPACK OK")
