#!/usr/bin/env python3
"""tests/golden/noise_stream.npz — NumPy's scan-noise stream, the golden for the device RNG.

The reference draws `rng.normal(0., 0.01, size=1080)` per scan from `np.random.default_rng(seed)`
(laser_models.py:450-452, base_classes.py:204; F110Env's default seed is 12345, f110_env.py:107).
numpy is a third-party dependency of the reference, so the vectors come from NumPy itself (no
reference code involved): for each seed the first and last rows of a 10^4-row stream, a SHA-256 of
all 10^4 x 1080 doubles, the positions of the first ziggurat tail / wedge samples, and the
generator state afterwards.  10^4 rows hold ~2800 tail samples and ~10^5 wedge tests per seed."""
import hashlib
import os

import numpy as np

SEEDS = [12345, 0, 2 ** 32 - 1]
ROWS, B, STD = 10000, 1080, 0.01
R = 3.6541528853610087963519472518


def main():
    out = {"seeds": np.array(SEEDS, dtype=np.uint64), "rows": ROWS, "beams": B, "std": STD, "numpy_version": np.__version__}
    for s in SEEDS:
        g = np.random.Generator(np.random.PCG64(s))
        x = g.normal(0., STD, size=(ROWS, B))
        st = g.bit_generator.state["state"]["state"]
        out["first_%d" % s] = x[:4].copy()
        out["last_%d" % s] = x[-2:].copy()
        out["sha256_%d" % s] = hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()
        out["state_%d" % s] = np.array([st >> 64, st & (2 ** 64 - 1)], dtype=np.uint64)
        tail = np.argwhere(np.abs(x) > STD * R)[:64]
        out["tail_pos_%d" % s] = tail.astype(np.int32)
        out["tail_val_%d" % s] = x[tail[:, 0], tail[:, 1]]
        out["row_sums_%d" % s] = x.sum(axis=1)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "noise_stream.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
