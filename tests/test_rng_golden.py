"""The device scan-noise generator (f1tenth_gym_amd/csrc/f110_rng.hpp) on the CPU: the header is
__host__ __device__, tests/host_harness runs the very same per-draw code, chain resolution and jump
constants lane by lane.  Checked against the committed NumPy golden (tests/golden/noise_stream.npz,
tools/gen_noise_golden.py) and against NumPy's live stream.  The -m gpu twin is in
tests/test_gpu_round2.py."""
import ctypes as C
import hashlib
import math

import numpy as np
import pytest

from _util import gold
from test_host_math import hh  # noqa: F401  (fixture: builds / loads the host harness)

_u64p = C.POINTER(C.c_uint64)
_dp = C.POINTER(C.c_double)


def _words(seed):
    st = np.random.PCG64(seed).state['state']
    m = (1 << 64) - 1
    return [st['state'] >> 64, st['state'] & m, st['inc'] >> 64, st['inc'] & m]


def _rows(hh, seed, rows, B, std=0.01):
    hh.hh_noise_rows.argtypes = [_u64p, C.c_double, C.c_int, C.c_int, _dp, _u64p]
    si = (C.c_uint64 * 4)(*_words(seed))
    out = np.empty((rows, B))
    so = (C.c_uint64 * 2)()
    hh.hh_noise_rows(si, std, rows, B, out.ctypes.data_as(_dp), so)
    return out, (int(so[0]) << 64) | int(so[1])


def test_seed_sequence_restatement(hh):
    """SeedSequence(seed) + pcg64_set_seed == np.random.PCG64(seed)"""
    hh.hh_pcg64_seed.argtypes = [C.c_uint64, _u64p]
    rng = np.random.default_rng(3)
    seeds = [0, 1, 12345, 2 ** 32 - 1, 2 ** 32, 2 ** 63, 2 ** 64 - 1] + [int(v) for v in rng.integers(0, 2 ** 63, 200)]
    for s in seeds:
        o = (C.c_uint64 * 4)()
        hh.hh_pcg64_seed(s, o)
        assert list(o) == _words(s), s


def test_log1p_restatement_is_glibc(hh):
    hh.hh_log1p.restype = C.c_double
    hh.hh_log1p.argtypes = [C.c_double]
    rng = np.random.default_rng(11)
    u = rng.random(200000)
    xs = np.concatenate([-u, -u ** 4, -(1.0 - u * 1e-7), -u * 1e-10, [-0.0, 0.0, -0.2928, -0.2929, -0.29290, -0.5, -0.9999999999999999]])
    for x in xs:
        assert hh.hh_log1p(float(x)) == math.log1p(float(x)), x


def test_stream_equals_golden_and_live_numpy(hh):
    g = gold("noise_stream")
    rows, B, std = int(g["rows"]), int(g["beams"]), float(g["std"])
    for s in [int(v) for v in g["seeds"]]:
        x, st = _rows(hh, s, rows, B, std)
        assert np.array_equal(x[:4], g["first_%d" % s]) and np.array_equal(x[-2:], g["last_%d" % s])
        assert hashlib.sha256(x.tobytes()).hexdigest() == str(g["sha256_%d" % s])
        tp = g["tail_pos_%d" % s]
        assert np.array_equal(x[tp[:, 0], tp[:, 1]], g["tail_val_%d" % s])   # ziggurat tail branch (log1p)
        assert st == (int(g["state_%d" % s][0]) << 64) | int(g["state_%d" % s][1])
        gen = np.random.Generator(np.random.PCG64(s))
        assert np.array_equal(x, gen.normal(0., std, size=(rows, B)))


@pytest.mark.parametrize("B", [1, 2, 63, 64, 65, 127, 271, 4096])
def test_stream_other_widths(hh, B):
    rows = max(3, 30000 // B)
    x, st = _rows(hh, 77 + B, rows, B, 0.5)
    gen = np.random.Generator(np.random.PCG64(77 + B))
    assert np.array_equal(x, gen.normal(0., 0.5, size=(rows, B)))
    assert st == gen.bit_generator.state['state']['state']
