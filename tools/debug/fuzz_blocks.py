"""tests/test_gpu_round4.py::test_env_blocks_fuzz_of_call_sequences over many more seeds: python tools/debug/fuzz_blocks.py 10 90"""
import sys, os, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import f1tenth_gym_amd as amd
import test_gpu_round4 as t
lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); bad = 0
for seed in range(lo, hi):
    try:
        t.test_env_blocks_fuzz_of_call_sequences(amd, seed)
    except AssertionError as ex:
        bad += 1; print("seed", seed, "FAILED", ex)
print("seeds %d..%d: %d failures, %.1f s" % (lo, hi - 1, bad, time.time() - t0))
