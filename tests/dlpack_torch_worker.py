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
print("DLPACK OK")
