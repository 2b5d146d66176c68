#!/bin/bash
# Round-3 GPU sessions (one gpurun call each; every phase is bounded):
#   gpurun --timeout 1500 -- 'bash tools/gpu_r3.sh test finalize probes tracks preroll'
#   gpurun --timeout 1500 -- 'bash tools/gpu_r3.sh bench prof pmc'
cd "${GRAFT_REPO_ROOT:-.}"
R="$PWD"; export TMPDIR=/tmp
OUT=$R/gpurun_out; mkdir -p $OUT
{ rocminfo | grep -E "Marketing Name|gfx9|Compute Unit" | head -8; nproc; lscpu | grep "Model name" | head -1; } > $OUT/box.txt 2>&1
python -c "import __graft_entry__ as g; print(g.build())" > $OUT/build.log 2>&1
H="--only-headline --steps 300 --warmup 30"
X="env F110_LIB_VARIANT=experimental"
line() { grep -h '^{' "$1" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('%-34s %8.2f M/s  %.4f ms/step  resets %d' % ('$2', d['value']/1e6, d['ms_per_step'], d['config']['env_resets_in_timed_region']))
"; }
for MODE in "$@"; do
case $MODE in
test)
  timeout 2400 python -m pytest tests -m gpu -q -rs --maxfail=10 --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
  tail -40 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log
  ;;
testfast)   # the product-library run only (the nested experimental-build run is the slow half)
  F110_NESTED_SUITE=1 timeout 1500 python -m pytest tests -m gpu -q -rs --maxfail=10 --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  tail -30 $OUT/pytest_gpu.log
  ;;
finalize)
  # A = 2 finalize: fixed lanes per agent vs the window loop flattened over the workgroup
  for n in 1024 4096 16384 65536; do for f in 0 1; do
    F110_EXP=finalize_flat=$f timeout 200 $X python bench.py $H --agents $n > $OUT/fin_n${n}_f${f}.log 2>&1; line $OUT/fin_n${n}_f${f}.log "agents $n finalize_flat $f"
  done; done
  for l in 8 16 64; do
    F110_EXP=finalize_flat=1,finalize_lanes=$l timeout 200 $X python bench.py $H --agents 65536 > $OUT/fin_flat_l$l.log 2>&1; line $OUT/fin_flat_l$l.log "65536 flat AG-by-lanes $l"
  done
  ;;
roles)
  for n in 1024 4096 16384 65536; do for r in 0 1; do
    F110_EXP=finalize_roles=$r timeout 200 $X python bench.py $H --agents $n > $OUT/roles_n${n}_r$r.log 2>&1; line $OUT/roles_n${n}_r$r.log "agents $n finalize_roles $r"
  done; done
  F110_EXP=finalize_roles=1,finalize_lanes=16 timeout 200 $X python bench.py $H --agents 65536 > $OUT/roles_l16.log 2>&1; line $OUT/roles_l16.log "65536 roles, AG 16"
  F110_EXP=finalize_roles=1,finalize_lanes=16 timeout 200 $X python bench.py $H --agents 4096 > $OUT/roles_4096_l16.log 2>&1; line $OUT/roles_4096_l16.log "4096 roles, AG 16"
  F110_EXP=finalize_roles=1,finalize_lanes=8 timeout 200 $X python bench.py $H --agents 16384 > $OUT/roles_16384_l8.log 2>&1; line $OUT/roles_16384_l8.log "16384 roles, AG 32"
  ;;
finalize2)
  # AG choice for the flattened finalize at the middle sizes, and the pair test inside the finalize kernel for
  # big batches WITHOUT the in-step re-seat (crashed / parked cars: windows grow)
  for n in 4096 16384; do for l in 8 16 64; do
    F110_EXP=finalize_lanes=$l timeout 200 $X python bench.py $H --agents $n > $OUT/fin2_n${n}_l$l.log 2>&1; line $OUT/fin2_n${n}_l$l.log "agents $n flat, lanes $l"
  done; done
  for pa in 0 1; do
    F110_EXP=pair_always=$pa timeout 200 $X python bench.py $H --agents 65536 --policy parked --no-reset --preroll 0 > $OUT/fin2_parked_pa$pa.log 2>&1; line $OUT/fin2_parked_pa$pa.log "65536 parked, pair_always $pa"
    F110_EXP=pair_always=$pa timeout 200 $X python bench.py $H --agents 65536 --no-reset --preroll 100 > $OUT/fin2_noreset_pa$pa.log 2>&1; line $OUT/fin2_noreset_pa$pa.log "65536 no reset, pair_always $pa"
    F110_EXP=pair_always=$pa timeout 200 $X python bench.py $H --agents 65536 --separate-reset > $OUT/fin2_sepreset_pa$pa.log 2>&1; line $OUT/fin2_sepreset_pa$pa.log "65536 separate reset, pair_always $pa"
  done
  ;;
ray)
  # configs[1]: the ray-level pass (last step's longest rays on waves of their own, scalar-path samples)
  for n in 1024 2048 4096 8192; do for r in 0 1; do
    F110_EXP=ray_pass=$r timeout 200 $X python bench.py $H --agents $n > $OUT/ray_n${n}_r$r.log 2>&1; line $OUT/ray_n${n}_r$r.log "agents $n ray_pass $r"
  done; done
  for thr in 16 32 48 96 128; do
    F110_EXP=ray_pass=1,ray_thr=$thr timeout 200 $X python bench.py $H --agents 4096 > $OUT/ray_thr$thr.log 2>&1; line $OUT/ray_thr$thr.log "4096 ray_thr $thr"
  done
  for w in 512 1024 4096 8192; do
    F110_EXP=ray_pass=1,ray_waves=$w timeout 200 $X python bench.py $H --agents 4096 > $OUT/ray_w$w.log 2>&1; line $OUT/ray_w$w.log "4096 ray_waves $w"
  done
  for tt in 24 48 200; do
    F110_EXP=ray_pass=1,task_thr=$tt timeout 200 $X python bench.py $H --agents 4096 > $OUT/ray_tt$tt.log 2>&1; line $OUT/ray_tt$tt.log "4096 ray_pass 1, task_thr $tt"
  done
  ;;
latency)
  # small batches are bound by the per-task latency chain (tools/debug/scan_timeline.py): sizes and the timeline
  timeout 600 python tools/debug/tpw_sweep.py 1024,2048,4096,8192,16384,65536 0 > $OUT/latency_sizes.txt 2>&1; cat $OUT/latency_sizes.txt
  timeout 300 $X python tools/debug/scan_timeline.py 4096 -1 2 > $OUT/scan_timeline_4096.txt 2>&1; tail -42 $OUT/scan_timeline_4096.txt
  ;;
lists)
  # longest-first list: threshold, capacity (tasks / div) and walking order, small batches
  echo "# experimental build, bench.py $H, F110_EXP as named" > $OUT/late_lists.txt
  for n in 4096 1024; do
    for cfgs in "task_thr=96" "task_thr=64" "task_thr=48,task_cap_div=8" "task_thr=32,task_cap_div=4" "task_thr=24,task_cap_div=2" "task_thr=96,task_rev=1" "task_thr=48,task_cap_div=8,task_rev=1" "task_thr=32,task_cap_div=4,task_rev=1" "task_thr=24,task_cap_div=2,task_rev=1" "task_thr=16,task_cap_div=2,task_rev=1"; do
      F110_EXP=$cfgs timeout 200 $X python bench.py $H --agents $n > $OUT/lists_tmp.log 2>&1; line $OUT/lists_tmp.log "agents $n $cfgs" | tee -a $OUT/late_lists.txt
    done
  done
  ;;
order)
  # longest-first window: does the list pay above 160 000 tasks now that both kernels run 8 waves per SIMD?
  for n in 8192 16384 32768 65536; do for o in 0 1; do
    F110_EXP=task_order=$o timeout 200 $X python bench.py $H --agents $n > $OUT/order_tmp.log 2>&1; line $OUT/order_tmp.log "agents $n task_order $o"
  done; done
  F110_EXP=task_order=1,task_thr=150 timeout 200 $X python bench.py $H --agents 65536 > $OUT/order_tmp.log 2>&1; line $OUT/order_tmp.log "agents 65536 task_order 1 thr 150"
  F110_EXP=task_order=1,task_thr=150 timeout 200 $X python bench.py $H --agents 16384 > $OUT/order_tmp.log 2>&1; line $OUT/order_tmp.log "agents 16384 task_order 1 thr 150"
  ;;
variants)
  # compile-time A/Bs of the PRODUCT library (built here, next to the tree's): no register hint on the scan kernels
  # (7 waves per SIMD for the longest-first and per-env-map kernels, the compiler's own allocation for the rest), and
  # no longest-first list at any size
  FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DF110_SRC_HASH=\"probe\""
  /opt/rocm/bin/hipcc $FL -DF110_SCAN_WAVES_EXPR=1 -DF110_DIRS_WAVES_EXPR=1 f1tenth_gym_amd/csrc/f110_hip.hip -o f1tenth_gym_amd/probe_a_no_register_hint.so > /dev/null 2>&1 &
  /opt/rocm/bin/hipcc $FL -DF110_TASK_ORDER_MAX_TASKS=0 f1tenth_gym_amd/csrc/f110_hip.hip -o f1tenth_gym_amd/probe_b_no_longest_first.so > /dev/null 2>&1 &
  wait
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  (probe_zz_tree = the product library as built from the tree)"
    echo "# bench.py --only-headline --steps 400 --warmup 20 --agents N"
    timeout 900 python tools/debug/lib_variants.py 1024,2048,4096,8192,16384,65536
    echo "# ... --agents 65536 --beams 4096 --map-tiles 2 --steps 60 --preroll 100 (BASELINE configs[4])"
    timeout 600 python tools/debug/lib_variants.py 65536 --beams 4096 --map-tiles 2 --steps 60 --preroll 100; } > $OUT/late_variants.txt 2>&1
  rm -f f1tenth_gym_amd/probe_*.so
  cat $OUT/late_variants.txt
  ;;
duo)
  # k_integrate in two waves per 64 agents (the low-speed branch's tan / cos one stage ahead) against one
  for n in 1024 4096 16384 65536; do for d in 0 1; do
    F110_EXP=integrate_duo=$d timeout 200 $X python bench.py $H --agents $n > $OUT/duo_tmp.log 2>&1; line $OUT/duo_tmp.log "agents $n integrate_duo $d" | tee -a $OUT/late_duo.txt
  done; done
  ;;
soak)
  # long-run identity of the product's dispatch against round 1's form, random-configuration fuzz against the oracle, VecEnv rate
  C=$(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')
  { echo "# round-3 soak and fuzz evidence, csrc $C"
    echo "## tools/debug/soak_round3.py 32768 3000 (experimental build): the product's dispatch vs round 1's form, 65 536 agents, 3000 steps"
    timeout 900 $X python tools/debug/soak_round3.py 32768 3000 2>&1 | tail -16
    echo "## tools/debug/fuzz_parity.py 0 250 (product build) — HIP vs oracle over random configurations: flags / step counters exact, floats |a-b| <= 1e-9*|b| + 1e-12"
    timeout 1200 python tools/debug/fuzz_parity.py 0 250 2>&1 | tail -2
    echo "## tools/debug/fuzz_parity.py 250 400 (experimental build, all five layouts)"
    timeout 1200 $X python tools/debug/fuzz_parity.py 250 400 2>&1 | tail -2
    echo "## tools/debug/fuzz_units.py 1 2 3 (product build): unit entry points vs oracle"
    timeout 600 python tools/debug/fuzz_units.py 1 2 3 2>&1 | tail -20; } > $OUT/soak_fuzz.txt 2>&1
  tail -30 $OUT/soak_fuzz.txt
  { echo "# csrc $C  tools/debug/vecenv_rate.py"; timeout 600 python tools/debug/vecenv_rate.py 2>&1 | tail -6; } > $OUT/vecenv_rate.txt 2>&1; cat $OUT/vecenv_rate.txt
  ;;
many)
  # envs of 1 / 3 / 4 agents (k_finalize_solo / k_finalize_multi) and, experimental build, round 1's form for 3 / 4
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  bench.py --only-headline --agents 65520|65536 --agents-per-env A --steps 200 --warmup 20"
    for a in 1 3 4 6 8 12 16; do
      n=65520; [ $a = 4 ] && n=65536; [ $a = 8 ] && n=65536; [ $a = 16 ] && n=65536
      timeout 300 python bench.py --only-headline --agents $n --agents-per-env $a --steps 200 --warmup 20 > $OUT/many_tmp.log 2>&1; line $OUT/many_tmp.log "A=$a product"
    done
    for a in 3 4 8 16; do
      n=65520; [ $a != 3 ] && n=65536
      F110_EXP=collide_mode=0 timeout 300 $X python bench.py --only-headline --agents $n --agents-per-env $a --steps 200 --warmup 20 > $OUT/many_tmp.log 2>&1; line $OUT/many_tmp.log "A=$a k_collide + k_finalize, round-1 form"
    done; } > $OUT/late_many_agents.txt 2>&1
  cat $OUT/late_many_agents.txt
  ;;
agsweep)
  # agents per workgroup of the A = 2 finalize kernel (finalize_lanes 8 / 16 / 64 -> AG 32 / 16 / 4), small batches
  for n in 1024 2048 4096 8192 16384; do for l in 8 16 64; do
    F110_EXP=finalize_lanes=$l timeout 200 $X python bench.py $H --agents $n > $OUT/ag_tmp.log 2>&1; line $OUT/ag_tmp.log "agents $n finalize_lanes $l" | tee -a $OUT/late_agsweep.txt
  done; done
  ;;
pmclegs)
  cd /tmp
  for cfg in "4096:--agents 4096" "cfg5:--agents 65536 --beams 4096 --map-tiles 2 --steps 100 --warmup 20 --preroll 100"; do
    tagc=${cfg%%:*}; argsc=${cfg#*:}; n=300; [ "$tagc" = cfg5 ] && n=100
    timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-include-regex "k_scan_rays|k_scan_dirs" -T -f csv -d $OUT/tr_vm -o p -- python $R/bench.py --only-headline $argsc > $OUT/tr_${tagc}_vm.log 2>&1
    python $R/tools/summarize_prof.py pmc $OUT/tr_vm $OUT/traffic_${tagc}_VMEM.json - $n
    rm -rf $OUT/tr_vm
  done
  cd "$R"; cat $OUT/traffic_4096_VMEM.json | head -30
  ;;
pmccfg5)
  cd /tmp
  i=0
  for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS" "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_TA_BUSY GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $ctrs --kernel-include-regex "k_scan_dirs" -T -f csv -d $OUT/pc5_$i -o p -- python $R/bench.py --only-headline --agents 65536 --beams 4096 --map-tiles 2 --steps 100 --warmup 20 --preroll 100 > $OUT/pc5_$i.log 2>&1
    python $R/tools/summarize_prof.py pmc $OUT/pc5_$i $OUT/pmc_cfg5_pass$i.json - 100
    rm -rf $OUT/pc5_$i
  done
  cd "$R"; python - <<'PYEOF'
import json, glob
m = {}
for f in sorted(glob.glob("gpurun_out/pmc_cfg5_pass*.json")):
    for k, r in json.load(open(f)).items():
        m.update(r["mean_per_dispatch"]); meta = r["meta"]
cyc = m["GRBM_GUI_ACTIVE"] / 8.0
tasks = 65536 * 24
print("kernel cycles %.0f (%.3f ms at 2.4 GHz)  VGPR %s SGPR %s" % (cyc, cyc / 2.4e6, meta.get("VGPR_Count"), meta.get("SGPR_Count")))
print("per task: VALU %.1f  SALU %.1f  VMEM rd %.2f wr %.2f  LDS %.2f" % (m["SQ_INSTS_VALU"] / tasks, m["SQ_INSTS_SALU"] / tasks, m["SQ_INSTS_VMEM_RD"] / tasks, m["SQ_INSTS_VMEM_WR"] / tasks, m["SQ_INSTS_LDS"] / tasks))
print("TA busy %.3f  TD busy %.3f  VALU active (SQ_ACTIVE_INST_VALU / 4 / waves-cycles) %.3f" % (m["TA_TA_BUSY_sum"] / 256.0 / cyc, m["TD_TD_BUSY_sum"] / 256.0 / cyc, m["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc)))
print("TCP hit %.3f  TCC hit %.3f  waves %d  wave cycles per wave %.0f" % (1 - m["TCP_TCC_READ_REQ_sum"] / m["TCP_TOTAL_CACHE_ACCESSES_sum"], m["TCC_HIT_sum"] / m["TCC_REQ_sum"], m["SQ_WAVES"], 4 * m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"]))
json.dump(m, open("gpurun_out/pmc_cfg5.json", "w"), indent=1, sort_keys=True)
PYEOF
  ;;
probes)
  # fusion feasibility: the scan kernel at the occupancy a fused (118-VGPR) kernel would have,
  # and with a per-env completion counter
  for n in 4096 65536; do
    F110_EXP=task_order=0 timeout 200 $X python bench.py $H --agents $n > $OUT/probe_n${n}_base.log 2>&1; line $OUT/probe_n${n}_base.log "agents $n base (no task order)"
    F110_EXP=task_order=0,scan_occupancy=4 timeout 200 $X python bench.py $H --agents $n > $OUT/probe_n${n}_occ4.log 2>&1; line $OUT/probe_n${n}_occ4.log "agents $n scan at 4 waves/SIMD"
    F110_EXP=task_order=0,scan_env_counter=1 timeout 200 $X python bench.py $H --agents $n > $OUT/probe_n${n}_cnt.log 2>&1; line $OUT/probe_n${n}_cnt.log "agents $n + per-env counter"
  done
  ;;
tracks)
  timeout 600 python tools/debug/track_scaling.py 65536 1080 > $OUT/track_scaling.log 2>&1; cat $OUT/track_scaling.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('tracks %2d %-18s %.4f ms/step' % (d['tracks'], d['assignment'], d['ms_per_step']))
    else: print(l.rstrip())"
  ;;
preroll)
  # how long until the batch is in its steady regime: 20 timed steps after P un-timed ones
  for p in 0 100 300 500 1000 2000; do
    timeout 200 python bench.py --only-headline --steps 20 --warmup 5 --preroll $p > $OUT/preroll_$p.log 2>&1; line $OUT/preroll_$p.log "preroll $p (20 steps)"
  done
  timeout 200 python bench.py --only-headline --steps 1000 --warmup 100 --preroll 0 > $OUT/preroll_steady.log 2>&1; line $OUT/preroll_steady.log "steady 100+1000"
  ;;
bench)
  timeout 900 python bench.py > $OUT/bench_default.log 2>&1; echo "bench exit $?" >> $OUT/bench_default.log
  tail -c 7000 $OUT/bench_default.log
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 --no-config5 --fixed-pose-steps 0 > $OUT/bench_driver_form.log 2>&1; line $OUT/bench_driver_form.log "driver form --steps 20 --warmup 5"
  ;;
prof)
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_stats -o stats -- python $R/bench.py $H > $OUT/prof_stats.log 2>&1
  python $R/tools/summarize_prof.py stats $OUT/prof_stats $OUT/kernel_stats.txt 300; rm -rf $OUT/prof_stats
  timeout 400 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_stats4k -o stats -- python $R/bench.py $H --agents 4096 > $OUT/prof_stats4k.log 2>&1
  python $R/tools/summarize_prof.py stats $OUT/prof_stats4k $OUT/kernel_stats_4096.txt 300; rm -rf $OUT/prof_stats4k
  cd "$R"; cat $OUT/kernel_stats.txt | head -30
  ;;
pmc)
  cd /tmp
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_TA_BUSY GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $ctrs --kernel-include-regex "k_scan_rays|k_finalize|k_integrate|k_collide" -T -f csv -d $OUT/pmc_$i -o p -- python $R/bench.py $H > $OUT/pmc_$i.log 2>&1
    python $R/tools/summarize_prof.py pmc $OUT/pmc_$i $OUT/pmc_pass$i.json - 300   # the 300 timed steps only (steady regime: after pre-roll + warm-up)
    rm -rf $OUT/pmc_$i
  done
  # HBM traffic of the scan kernel for the two other bench legs (FETCH_SIZE, WRITE_SIZE)
  for cfg in "4096:--agents 4096" "cfg5:--agents 65536 --beams 4096 --map-tiles 2 --steps 100 --warmup 20 --preroll 100"; do
    tagc=${cfg%%:*}; argsc=${cfg#*:}; n=300; [ "$tagc" = cfg5 ] && n=100
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --pmc $c --kernel-include-regex "k_scan_rays|k_scan_dirs" -T -f csv -d $OUT/tr_$c -o p -- python $R/bench.py --only-headline $argsc > $OUT/tr_${tagc}_$c.log 2>&1
      python $R/tools/summarize_prof.py pmc $OUT/tr_$c $OUT/traffic_${tagc}_$c.json - $n
      rm -rf $OUT/tr_$c
    done
    # ... and its wave-level vector-memory instructions (the gather-issue floor of that leg)
    timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-include-regex "k_scan_rays|k_scan_dirs" -T -f csv -d $OUT/tr_vm -o p -- python $R/bench.py --only-headline $argsc > $OUT/tr_${tagc}_vm.log 2>&1
    python $R/tools/summarize_prof.py pmc $OUT/tr_vm $OUT/traffic_${tagc}_VMEM.json - $n
    rm -rf $OUT/tr_vm
  done
  cd "$R"
  ;;
esac
done
