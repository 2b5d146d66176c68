#!/bin/bash
# Round-6 GPU sessions (one gpurun call each; EVERY command under its own timeout — a hung rocprofv3 cost 20 GPU-minutes once):
#   gpurun --timeout 1500 -- 'bash tools/gpu_r6.sh test stream'
# modes: test new testfast bench prof pmc tabench rates manyagents | round 6: cfg3 (BASELINE configs[3] at its size, 8 ranks on this one device, RCCL stand-in) fuzztime envprof
#   (the modes of round 6's retired experiments — pair pairsweep ntstore pairpmc finwaves — went with their switches: commit 'row-pair experiment measured' has them; outputs: profiles/r06_rowpair*.txt, r06_nt_store.txt, r06_finalize_waves.txt)
cd "${GRAFT_REPO_ROOT:-.}"
R="$PWD"; export TMPDIR=/tmp
OUT=$R/gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; print(g.build())" > $OUT/build.log 2>&1
{ rocminfo | grep -E "Marketing Name|gfx9|Compute Unit|Max Clock" | head -8; nproc; lscpu | grep -E "Model name|Socket|NUMA node\(s\)|Core\(s\) per socket" | head -4; } > $OUT/box.txt 2>&1
H="--only-headline --steps 300 --warmup 30"
C5="--only-headline --agents 65536 --beams 4096 --map-tiles 2 --steps 100 --warmup 20 --preroll 100"
line() { grep -h '^{' "$1" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('%-34s %8.2f M/s  %.4f ms/step  resets %d' % ('$2', d['value']/1e6, d['ms_per_step'], d['config']['env_resets_in_timed_region']))
"; }
for MODE in "$@"; do
case $MODE in
test)
  timeout 1500 python -m pytest tests -m gpu -q -rs --maxfail=10 --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
  tail -45 $OUT/pytest_gpu.log | cut -c1-220; tail -2 $OUT/smoke.log
  ;;
new)   # this round's test file only
  timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py::test_dlpack_hand_off_to_torch tests/test_gpu_round5.py::test_example_rl_loop_device_runs -m gpu -q -rs --maxfail=20 --durations=15 > $OUT/pytest_new.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_new.log
  tail -60 $OUT/pytest_new.log | cut -c1-300
  ;;
testfast)
  F110_NESTED_SUITE=1 timeout 900 python -m pytest tests -m gpu -q -x --durations=10 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  tail -16 $OUT/pytest_gpu.log | cut -c1-220
  ;;
cfg3)   # BASELINE configs[3] at its real size on ONE device: 8 ranks x 32 768 agents as threads of one process, tests/rccl_stub standing in for RCCL
  python - <<'PY' > /dev/null 2>&1
import os, shutil, subprocess
d = "tests/rccl_stub"; lib, src = d + "/librccl.so.1", d + "/rccl_stub.hip"
if not os.path.isfile(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
    subprocess.check_call([shutil.which("hipcc") or "/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", lib])
PY
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  LD_LIBRARY_PATH=tests/rccl_stub F110_BENCH_DEVICE=0 python bench.py --gpus 8 --ranks-in-process --agents 32768 --steps 20 --warmup 5"
    echo "# 8 ranks = 8 threads x 1 handle, ALL on device 0 (a correctness / capacity run of the N-rank code at configs[3]'s size, not a scaling claim: the ranks share one GPU and the 'links' are device-to-device copies)"
    ( time LD_LIBRARY_PATH=$R/tests/rccl_stub:$LD_LIBRARY_PATH F110_BENCH_DEVICE=0 timeout 900 python bench.py --gpus 8 --ranks-in-process --agents 32768 --steps 20 --warmup 5 --gather-timeout 300 --gather-budget 600 > $OUT/cfg3_bench.log 2>$OUT/cfg3_bench.err ) 2> $OUT/cfg3_bench.time
    grep -h '^{' $OUT/cfg3_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); mg = d['multi_gpu']
print('headline (no collective)        %8.2f M agent-steps/s  %.4f ms/step  per-rank ms %s  agents_total %d' % (d['value']/1e6, d['ms_per_step'], ['%.3f' % x for x in mg['per_rank_ms_per_step']], d['config']['agents_total']))
for leg in ('gather', 'gather_overlap', 'gather_f32', 'gather_f32_overlap', 'gather_root', 'gather_root_f32_overlap'):
    r = mg.get(leg)
    if r: print('%-31s %8.2f M agent-steps/s  %.4f ms/step  gather_ok %s  rccl_ranks %s  device memory in use %.1f GB  bytes received per step %s' % (leg, r['value']/1e6, r['ms_per_step'], r['gather_ok'], r['rccl_ranks'], r.get('device_mem_used_gb_max', -1), r['bytes_received_per_step']))
print('legs_skipped', mg.get('legs_skipped'), 'gather_error', mg.get('gather_error'), '| ranks are', mg.get('ranks_are'))
"; tail -3 $OUT/cfg3_bench.time
    echo "# tests/rccl_stub/config3_full_size.py 8 16384 (digests of every peer block, twins across block boundaries, 32 envs per rank vs the oracle)"
    LD_LIBRARY_PATH=$R/tests/rccl_stub:$LD_LIBRARY_PATH timeout 900 python tests/rccl_stub/config3_full_size.py 8 16384 2>&1 | grep RESULT; } | tee $OUT/cfg3_one_device.txt
  ;;
fuzztime)   # how long the fuzzers take per seed on this box (sizes the in-suite seed counts)
  { TIMEFORMAT="%R s"
    for f in fuzz_parity fuzz_envs fuzz_episode; do echo -n "$f seeds 0..39: "; { time timeout 600 python tools/debug/$f.py 0 40 > $OUT/ft_$f.log 2>&1; } 2>&1; tail -1 $OUT/ft_$f.log | cut -c1-200; done
    echo -n "fuzz_units seeds 0..3: "; { time bash -c 'for s in 0 1 2 3; do timeout 300 python tools/debug/fuzz_units.py $s > /dev/null 2>&1 || echo "fuzz_units seed $s FAILED"; done'; } 2>&1; } | tee $OUT/fuzztime.txt
  ;;
fuzzlong)   # the fuzzers by hand, far beyond the seeds the suite runs (supplementary evidence: profiles/r06_fuzz.txt)
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  the GPU fuzzers on seed ranges BEYOND the suite's (HIP path vs CPU oracle / host logic; a line per fuzzer: seeds, ok count, failed seeds)"
    for spec in "fuzz_host 200 1400" "fuzz_parity 300 1500" "fuzz_envs 200 800" "fuzz_episode 300 1500"; do
      set -- $spec
      timeout 1200 python tools/debug/$1.py $2 $3 > $OUT/fl_$1.log 2>&1
      echo "$1 seeds $2..$(( $3 - 1 )): $(grep -c '^ok' $OUT/fl_$1.log) ok, $(grep -c MISMATCH $OUT/fl_$1.log) mismatches, $(tail -1 $OUT/fl_$1.log)"
      [ "$1" = fuzz_host ] && echo "   of which one launch per step (k_step_tiny): $(grep -c 'one launch per step' $OUT/fl_$1.log)"
    done
    bad=0; for sd in $(seq 12 41); do timeout 300 python tools/debug/fuzz_units.py $sd > $OUT/fl_units.log 2>&1 || bad=$((bad+1)); grep -q "False" $OUT/fl_units.log && bad=$((bad+1)); done
    echo "fuzz_units seeds 12..41: $bad seed(s) with a failure or an inexact line"; } | tee $OUT/fuzz_long.txt
  ;;
fuzzmore)   # ... and a second, disjoint set of seed ranges (appended to the same record)
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  a second set of seed ranges"
    for spec in "fuzz_host 1400 3000" "fuzz_parity 1500 3000" "fuzz_envs 800 1600" "fuzz_episode 1500 3000"; do
      set -- $spec
      timeout 1500 python tools/debug/$1.py $2 $3 > $OUT/fm_$1.log 2>&1
      echo "$1 seeds $2..$(( $3 - 1 )): $(grep -c '^ok' $OUT/fm_$1.log) ok, $(grep -c MISMATCH $OUT/fm_$1.log) mismatches, $(tail -1 $OUT/fm_$1.log)"
      [ "$1" = fuzz_host ] && echo "   of which one launch per step (k_step_tiny): $(grep -c 'one launch per step' $OUT/fm_$1.log)"
    done; } | tee $OUT/fuzz_more.txt
  ;;
soak)   # long auto-reset loops on the final sources: does memory stay flat, do the forms keep agreeing with themselves?
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  tools/debug/long_run_memory.py 300000 E (F110VecEnv device_logic auto_reset, noise from a 64-row cache: episodes beyond it continue from the carried stream state)"
    echo "## 16 envs x 2 (the per-kernel host path)"; timeout 600 python tools/debug/long_run_memory.py 300000 16 2>&1 | tail -13
    echo "## 2 envs x 2 (4 agents: k_step_tiny for the first 64 steps, then — the upper bound of any live episode's age has outgrown the 64-row cache and nothing resets it short of a full reset — the per-kernel form with k_noise_rows)"; timeout 600 python tools/debug/long_run_memory.py 300000 2 2>&1 | tail -13
    echo "## F110Env against the oracle at every step, 400 episodes, row cache of 64: the forms switch in mid-episode and at reset()"; timeout 900 python tools/debug/f110env_soak.py 400 2 2>&1 | tail -2; timeout 900 python tools/debug/f110env_soak.py 400 1 2>&1 | tail -2
    echo "## ... and with the default cache (rows ahead of need, doublings from 256), cars that crawl: long episodes through the doublings, later episodes replaying the rows"; timeout 900 python tools/debug/f110env_soak.py 40 2 0 6000 1 2>&1 | tail -2; timeout 900 python tools/debug/f110env_soak.py 40 1 0 6000 1 2>&1 | tail -2
    echo "## ShardedVecEnv: 4096 envs x 2 over four handles on device 0, 20 000 steps"; timeout 900 python examples/sharded_vec_env.py --envs 4096 --devices 0,0,0,0 --steps 20000 2>&1 | tail -2; } | tee $OUT/soak.txt
  ;;
tiny)   # k_step_tiny: the A/B tests (lab build) + the golden / oracle tests that now run through it (product), then F110Env's step time
  F110_LIB_VARIANT=experimental F110_NESTED_SUITE=1 timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -x -k "tiny or one_launch" > $OUT/pytest_tiny_lab.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_tiny_lab.log; tail -15 $OUT/pytest_tiny_lab.log | cut -c1-250
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round4.py -m gpu -q -x --deselect tests/test_gpu_round2.py::test_fuzz_parity_bounded_seeds --deselect tests/test_gpu_round2.py::test_fuzz_units_seeds > $OUT/pytest_tiny_product.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_tiny_product.log; tail -8 $OUT/pytest_tiny_product.log | cut -c1-250
  for i in 1 2 3; do timeout 120 python tools/debug/f110env_loop.py 3000 2>&1 | tail -1; done
  ;;
latency)   # where a host-synchronised step of ONE env spends its time: the launch floor, k_step_tiny's phases, the host's view, the first episode
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')"
    echo "## tools/debug/launch_latency2.hip: one launch + a completion word in page-locked memory, polled (no library of ours involved)"
    hipcc --offload-arch=gfx950 -O2 -o /tmp/ll2 tools/debug/launch_latency2.hip 2>/dev/null && timeout 120 /tmp/ll2
    echo "## tools/debug/tiny_timeline.py (lab build: phase stamps of the 100 MHz clock inside k_step_tiny)"
    for a in 2 1; do F110_LIB_VARIANT=experimental timeout 300 python tools/debug/tiny_timeline.py $a 2000; done
    echo "## tools/debug/tiny_start_probe.py (lab build: the first workgroup tells the host when it starts)"
    for a in 2 1; do F110_LIB_VARIANT=experimental timeout 300 python tools/debug/tiny_start_probe.py $a 4000; done
    echo "## tools/debug/tiny_drift.py (product build): BatchSim.step_host per block of 1000 steps from a fresh handle — the noise rows of a first episode are generated ahead of need"
    for a in 2 1; do timeout 300 python tools/debug/tiny_drift.py $a 8; done
    echo "## tools/debug/f110env_loop.py 3000 (product build, a fresh F110Env each)"
    for i in 1 2 3; do timeout 120 python tools/debug/f110env_loop.py 3000 2>&1 | tail -1; done; } 2>&1 | tee $OUT/launch_latency.txt
  ;;
tinyab)   # one launch vs three kernels on the shapes k_step_tiny serves (lab build carries the switch)
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  F110_LIB_VARIANT=experimental python tools/debug/tiny_ab.py"
    F110_LIB_VARIANT=experimental timeout 600 python tools/debug/tiny_ab.py 2>&1 | tail -6; } | tee $OUT/tiny_ab.txt
  ;;
envprof)   # where F110Env(num_agents=2).step's time goes: per-kernel durations (rocprofv3), host enqueue / wait, Python around the call
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  tools/debug/f110env_loop.py 3000 (F110Env 1 env x 2 agents)"
    for i in 1 2 3; do timeout 120 python tools/debug/f110env_loop.py 3000 2>&1 | tail -1; done
    timeout 120 python tools/debug/idle_gap_probe.py 2>&1 | tail -12
    cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_env -o stats -- python $R/tools/debug/f110env_loop.py 3000 > $OUT/prof_env.log 2>&1
    timeout 60 python $R/tools/summarize_prof.py stats $OUT/prof_env $OUT/kernel_stats_f110env.txt 3000; rm -rf $OUT/prof_env; cd "$R"
    cat $OUT/kernel_stats_f110env.txt | head -30; } | tee $OUT/envprof.txt
  ;;
manyagents)
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  bench.py --only-headline --agents 65520|65536 --agents-per-env A --steps 200 --warmup 20 (product library)"
  for a in 1 2 3 4 8 16 24 32; do
    n=$(( 65536 / a * a ))
    timeout 200 python bench.py --only-headline --agents $n --agents-per-env $a --steps 200 --warmup 20 > $OUT/many_$a.log 2>&1; line $OUT/many_$a.log "A=$a product"
  done; } | tee $OUT/many_agents.txt
  ;;
rates)
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  bench.py $H --agents N (product library)"
  for n in 1024 2048 4096 8192 16384 32768 65536; do timeout 200 python bench.py $H --agents $n > $OUT/rate_$n.log 2>&1; line $OUT/rate_$n.log "agents $n"; done; } | tee $OUT/rates.txt
  ;;
bench)
  ( time timeout 900 python bench.py > $OUT/bench_default.log 2>$OUT/bench_default.err ) 2> $OUT/bench_default.time; echo "bench exit $?" >> $OUT/bench_default.log; tail -3 $OUT/bench_default.time
  tail -c 2500 $OUT/bench_default.log
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --secondary 0 --no-config5 --fixed-pose-steps 0 > $OUT/bench_driver_form.log 2>&1; line $OUT/bench_driver_form.log "driver form --steps 20 --warmup 5"
  ;;
prof)
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_stats -o stats -- python $R/bench.py $H > $OUT/prof_stats.log 2>&1
  timeout 60 python $R/tools/summarize_prof.py stats $OUT/prof_stats $OUT/kernel_stats.txt 300; rm -rf $OUT/prof_stats
  timeout 400 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_stats4k -o stats -- python $R/bench.py $H --agents 4096 --groups 1 > $OUT/prof_stats4k.log 2>&1
  timeout 60 python $R/tools/summarize_prof.py stats $OUT/prof_stats4k $OUT/kernel_stats_4096.txt 300; rm -rf $OUT/prof_stats4k
  timeout 400 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_stats5 -o stats -- python $R/bench.py $C5 > $OUT/prof_stats5.log 2>&1
  timeout 60 python $R/tools/summarize_prof.py stats $OUT/prof_stats5 $OUT/kernel_stats_cfg5.txt 100; rm -rf $OUT/prof_stats5
  cd "$R"; head -14 $OUT/kernel_stats.txt; tail -6 $OUT/kernel_stats_4096.txt; tail -6 $OUT/kernel_stats_cfg5.txt
  ;;
pmc)
  cd /tmp
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_TA_BUSY GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $ctrs --kernel-include-regex "k_scan_rays|k_finalize|k_integrate|k_collide" -T -f csv -d $OUT/pmc_$i -o p -- python $R/bench.py $H > $OUT/pmc_$i.log 2>&1
    timeout 60 python $R/tools/summarize_prof.py pmc $OUT/pmc_$i $OUT/pmc_pass$i.json - 300
    rm -rf $OUT/pmc_$i
  done
  for cfg in "4096:--agents 4096 --groups 1" "cfg5:--agents 65536 --beams 4096 --map-tiles 2 --steps 100 --warmup 20 --preroll 100"; do
    tagc=${cfg%%:*}; argsc=${cfg#*:}; n=300; [ "$tagc" = cfg5 ] && n=100
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --pmc $c --kernel-include-regex "k_scan_rays|k_scan_dirs" -T -f csv -d $OUT/tr_$c -o p -- python $R/bench.py --only-headline $argsc > $OUT/tr_${tagc}_$c.log 2>&1
      timeout 60 python $R/tools/summarize_prof.py pmc $OUT/tr_$c $OUT/traffic_${tagc}_$c.json - $n
      rm -rf $OUT/tr_$c
    done
    timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-include-regex "k_scan_rays|k_scan_dirs" -T -f csv -d $OUT/tr_vm -o p -- python $R/bench.py --only-headline $argsc > $OUT/tr_${tagc}_vm.log 2>&1
    timeout 60 python $R/tools/summarize_prof.py pmc $OUT/tr_vm $OUT/traffic_${tagc}_VMEM.json - $n
    rm -rf $OUT/tr_vm
  done
  cd "$R"; ls $OUT/pmc_pass*.json $OUT/traffic_*.json 2>/dev/null | wc -l
  ;;
tabench)
  { echo "# $(date -u) tools/debug/ta_bench.hip on this box"; cat $OUT/box.txt; } > $OUT/ta_bench.txt
  timeout 120 hipcc --offload-arch=gfx950 -O3 tools/debug/ta_bench.hip -o /tmp/ta_bench > /dev/null 2>&1 && timeout 120 /tmp/ta_bench >> $OUT/ta_bench.txt 2>&1; tail -12 $OUT/ta_bench.txt
  ;;
esac
done
