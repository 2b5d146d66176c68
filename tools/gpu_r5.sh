#!/bin/bash
# Round-5 GPU sessions (one gpurun call each; EVERY command under its own timeout — a hung rocprofv3 cost 20 GPU-minutes once):
#   gpurun --timeout 1500 -- 'bash tools/gpu_r5.sh test stream'
# modes: test testfast stream bench prof
cd "${GRAFT_REPO_ROOT:-.}"
R="$PWD"; export TMPDIR=/tmp
OUT=$R/gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; print(g.build())" > $OUT/build.log 2>&1
for MODE in "$@"; do
case $MODE in
test)
  timeout 1500 python -m pytest tests -m gpu -q -rs --maxfail=10 --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
  tail -45 $OUT/pytest_gpu.log | cut -c1-220; tail -2 $OUT/smoke.log
  ;;
testfast)
  F110_NESTED_SUITE=1 timeout 900 python -m pytest tests -m gpu -q -x --durations=10 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  tail -16 $OUT/pytest_gpu.log | cut -c1-220
  ;;
stream)   # the lane-refill scan against k_scan_rays_agent (experimental build), bench-like loop with in-step re-seats
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  tools/debug/stream_probe.py N key=value ... (experimental build; 300 pre-roll + 100 timed steps, in-step re-seats)"
    P="timeout 90 env F110_LIB_VARIANT=experimental python tools/debug/stream_probe.py"
    for n in 65536 16384; do
      $P $n scan_stream=0
      for r in 64 56 48 32 16; do $P $n scan_stream=1 stream_refill=$r; done
      for r in 64 48 32 16; do $P $n scan_stream=1 stream_block=64 stream_refill=$r; done
      $P $n scan_stream=1 stream_grid=512 stream_refill=48
      $P $n scan_stream=1 stream_block=256 stream_grid=2048 stream_refill=48
    done 2>&1 | grep agents; } | tee $OUT/stream_scan.txt
  ;;
spec)   # march_padded_spec (two samples per round trip in a long ray's tail) in the longest-first window, experimental build
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  F110_LIB_VARIANT=experimental F110_EXP=spec_from=S python bench.py --only-headline --steps 300 --warmup 30 --agents N (spec_from 0 = the plain march)"
    for n in 2048 4096 8192; do for sp in 0 64 32 16 8 4; do
      F110_LIB_VARIANT=experimental F110_EXP=spec_from=$sp timeout 100 python bench.py --only-headline --steps 300 --warmup 30 --agents $n 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('agents %6d spec_from %2d  %8.2f M agent-steps/s  %.4f ms/step' % ($n, $sp, d['value']/1e6, d['ms_per_step']))
"; done; done; } | tee $OUT/spec_march.txt
  ;;
finwave)   # the A = 2 finalize as one-wave workgroups (k_finalize_pair_roles<AG, 64>) under one / two env blocks, experimental build
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  F110_LIB_VARIANT=experimental F110_EXP=finalize_wave=W python bench.py --only-headline --steps 300 --warmup 30 --agents N --groups G (W 0 = the 256-thread product form)"
    for n in 65536 32768; do for g in 1 2; do for fw in 0 8 4; do
      F110_LIB_VARIANT=experimental F110_EXP=finalize_wave=$fw timeout 100 python bench.py --only-headline --steps 300 --warmup 30 --agents $n --groups $g 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('agents %6d env blocks %d finalize_wave %d  %8.2f M agent-steps/s  %.4f ms/step' % ($n, $g, $fw, d['value']/1e6, d['ms_per_step']))
"; done; done; done; } | tee $OUT/finalize_wave.txt
  ;;
bench)
  timeout 900 python bench.py > $OUT/bench_default.log 2>&1; grep -h '^{' $OUT/bench_default.log | tail -1 > $OUT/bench_default.json; cut -c1-600 $OUT/bench_default.json
  ;;
esac
done
