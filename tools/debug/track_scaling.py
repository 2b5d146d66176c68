"""SURVEY 8(f)-2 / VERDICT r2 #3: what many distinct tracks cost.  65 536 agents; 1 / 2 / 4 / 8 / 16 registered
tracks (copies of example_map: distinct tables in HBM, identical content so the bench's start poses are valid on
every one), assigned to the envs interleaved (env e -> track e mod k: every XCD's L2 sees every track) or
grouped (contiguous env ranges per track: the scan's XCD-contiguous block order then gives each XCD's L2 one
eighth of the envs, i.e. k / 8 tracks).  Prints one JSON line per configuration and writes
gpurun_out/track_scaling.json (copied to profiles/ by tools/collect_profiles.py)."""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import f1tenth_gym_amd as amd  # noqa: E402
from f1tenth_gym_amd import build  # noqa: E402
from _util import load_map_image, bench_start_poses  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
BEAMS = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
MAX_TRACKS = 16
img, res, origin = load_map_image("example_map")
E, A = N // 2, 2
s = amd.BatchSim(num_envs=E, num_agents=A, num_beams=BEAMS)
s.set_map_image(img, res, origin)
for m in range(1, MAX_TRACKS):
    assert s.add_map_image(img, res, origin) == m
s.set_noise_rng(12345, 0.01)
s.noise_prepare(400)
poses = bench_start_poses(E, A)
rng = np.random.default_rng(0)
d_act = s.device_array((E * A, 2)); d_act.upload(np.stack([rng.uniform(-0.2, 0.2, E * A), rng.uniform(2, 6, E * A)], axis=1))
d_start = s.device_array((E * A, 3)); d_start.upload(poses)
out = []
free, total = s.device_mem_info()
# two sweeps, the second in reverse order (clock / thermal state drifts over a run); the faster of the two counts
sweep = [(k, a) for k in (0, 1, 2, 4, 8, 16) for a in (("single-map kernel",) if k == 0 else (("one slot",) if k == 1 else ("interleaved", "grouped")))]
best = {}
for k, assign in sweep + sweep[::-1]:
    if True:
        if k == 0:
            s.set_env_maps(None)
        elif assign == "grouped":
            s.set_env_maps((np.arange(E) * k) // E)
        else:
            s.set_env_maps(np.arange(E) % k)
        s.set_auto_reseat(None); s.reset_device(d_start); s.set_auto_reseat(d_start, 0, None)
        for _ in range(100):
            s.step_device(d_act)
        s.sync(); t0 = time.perf_counter()
        for _ in range(150):
            s.step_device(d_act)
        s.sync(); dt = (time.perf_counter() - t0) / 150
        rec = {"agents": N, "beams": BEAMS, "tracks": max(k, 1), "assignment": assign, "ms_per_step": dt * 1e3, "agent_steps_per_s": N / dt,
               "csrc": build.src_hash()}
        if (k, assign) not in best or rec["ms_per_step"] < best[(k, assign)]["ms_per_step"]:
            best[(k, assign)] = rec
out = [best[c] for c in sweep]
for rec in out:
    print(json.dumps(rec)); sys.stdout.flush()
base = out[0]["ms_per_step"]
for r in out:
    r["vs_single_map"] = r["ms_per_step"] / base
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"device_bytes_in_use_with_16_tracks": total - free, "table": out}, open(os.path.join(ROOT, "gpurun_out", "track_scaling_%d_%d.json" % (N, BEAMS)), "w"), indent=1)
s.close()
