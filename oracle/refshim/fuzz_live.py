"""LIVE reference against the oracle on RANDOM configurations (build container only; test infrastructure).

tests/test_reference_fuzz.py walks 25 hand-written configurations.  This script draws configurations at random — track,
resolution / origin / yaw overrides, beams, fov, eps, theta_dis, max_range for `ScanSimulator2D`; cars, integrator, time step,
lidar offset, steps for `Simulator`; the same plus ego index for `F110Env` — adds them to that module's case tables and
runs its three checks on each (the reference imported from /root/reference through ref_loader, the oracle beside it).
    python oracle/refshim/fuzz_live.py 0 100        # seeds 0..99: one scan, one simulator and (every third seed) one env case each
"""
import os
import sys
import tempfile
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

TRACKS = ["example_map", "berlin", "skirk", "vegas", "stata_basement"]


def draw_map(rng):
    name = str(rng.choice(TRACKS))
    if rng.random() < 0.35:     # another resolution, origin and yaw on the same image (laser_models.py:55-86, :417-420)
        res = float(rng.choice([0.05, 0.07, 0.0437, 0.11, 0.0625]))
        org = [float(rng.uniform(-40, 5)), float(rng.uniform(-30, 5)), float(rng.choice([0.0, 0.3, -1.1, 2.4, 3.0]))]
        return name, res, org
    return name, None, None


def main(first, last):
    import test_reference_fuzz as T
    if not T.ref_loader.reference_available():
        print("the reference tree is not here: nothing to do"); return 1
    bad = []
    for tab in ("SCAN_CASES", "SIM_CASES", "ENV_CASES"):      # index -> case, so that a seed's case keeps its index whatever ran before it
        setattr(T, tab, dict(enumerate(getattr(T, tab))))
    for seed in range(first, last):
        rng = np.random.default_rng(300000 + seed)
        name, res, org = draw_map(rng)
        scan = (name, res, org, int(rng.choice([64, 100, 180, 271, 540, 1080, 1500])), float(rng.choice([4.7, 4.7, 3.0, 6.0, 6.28])),
                float(rng.choice([1e-4, 1e-4, 0.03, 0.2])), int(rng.choice([2000, 2000, 720, 1000, 3600])), float(rng.choice([30.0, 30.0, 8.0, 12.5])))
        name2, res2, org2 = draw_map(rng)
        sim = (name2, res2, org2, int(rng.integers(1, 6)), str(rng.choice(["RK4", "RK4", "Euler"])), float(rng.choice([0.01, 0.01, 0.005, 0.02])),
               float(rng.choice([0.0, 0.0, 0.1, 0.275])), int(rng.integers(15, 40)))
        env = (str(rng.choice(TRACKS[:3])) if rng.random() < 0.8 else None, int(rng.integers(1, 4)), 0, str(rng.choice(["RK4", "Euler"])),
               float(rng.choice([0.01, 0.02])), float(rng.choice([0.0, 0.1, 0.275])), int(rng.integers(60, 200)))
        env = env[:2] + (int(rng.integers(0, env[1])),) + env[3:]
        if rng.random() < 0.5:      # laps by circling forwards at full lock instead of back and forth through the start zone
            env = env[:6] + (int(rng.integers(150, 330)), "circle")
        jobs = [("scan", T.SCAN_CASES, scan, T.test_scan_simulator_ctor_and_map_sweep, True),
                ("sim", T.SIM_CASES, sim, T.test_simulator_sweep, True)]
        if seed % 3 == 0:
            jobs.append(("env", T.ENV_CASES, env, T.test_f110env_sweep, False))
        for kind, table, case, fn, wants_tmp in jobs:
            idx = 1000 + seed; table[idx] = case      # (the checks seed their own generators with the case index)
            try:
                with tempfile.TemporaryDirectory() as tmp:
                    fn(idx, tmp) if wants_tmp else fn(idx)
                print("ok seed %d %s %s" % (seed, kind, case), flush=True)
            except Exception as ex:     # noqa: BLE001 — report and go on
                bad.append((seed, kind))
                print("MISMATCH seed %d %s %s: %s" % (seed, kind, case, "".join(traceback.format_exception_only(type(ex), ex)).strip()[:600]), flush=True)
    print("failed:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]), int(sys.argv[2])))
