#!/bin/bash
# Round-4 GPU sessions (one gpurun call each; every phase is bounded):
#   gpurun --timeout 3000 -- 'bash tools/gpu_r4.sh test gatherlegs'                      # both builds, smoke, the gather legs at world size 1
#   gpurun --timeout 3400 -- 'bash tools/gpu_r4.sh prof pmc pmccfg5 pmc4096 tabench manyagents dropin bench'   # everything profiles/r04_* holds
#   then: python tools/collect_profiles.py r04 65536 1080 3
# modes: test testfast rates cfg5 manyagents fan dropin gatherlegs bench prof pmc pmc4096 pmccfg5 tabench
cd "${GRAFT_REPO_ROOT:-.}"
R="$PWD"; export TMPDIR=/tmp
OUT=$R/gpurun_out; mkdir -p $OUT
{ rocminfo | grep -E "Marketing Name|gfx9|Compute Unit|Max Clock" | head -8; nproc; lscpu | grep -E "Model name|Socket|NUMA node\(s\)|Core\(s\) per socket" | head -4; } > $OUT/box.txt 2>&1
python -c "import __graft_entry__ as g; print(g.build())" > $OUT/build.log 2>&1
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
  timeout 2400 python -m pytest tests -m gpu -q -rs --maxfail=10 --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
  tail -40 $OUT/pytest_gpu.log | cut -c1-220; tail -2 $OUT/smoke.log
  ;;
testfast)   # the product-library run only (the nested experimental-build run is the slow half)
  F110_NESTED_SUITE=1 timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  tail -16 $OUT/pytest_gpu.log | cut -c1-220
  ;;
rates)
  for n in 1024 2048 4096 8192 16384 65536; do timeout 200 python bench.py $H --agents $n > $OUT/rate_$n.log 2>&1; line $OUT/rate_$n.log "agents $n"; done | tee $OUT/rates.txt
  for i in 1 2; do timeout 300 python bench.py $C5 > $OUT/rate_cfg5.log 2>&1; line $OUT/rate_cfg5.log "configs[4]"; done | tee -a $OUT/rates.txt
  ;;
cfg5)
  for i in 1 2; do timeout 300 python bench.py $C5 > $OUT/rate_cfg5.log 2>&1; line $OUT/rate_cfg5.log "configs[4]"; done | tee $OUT/rates_cfg5.txt
  ;;
manyagents)
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  bench.py --only-headline --agents 65520|65536 --agents-per-env A --steps 200 --warmup 20"
  for a in 1 2 3 4 6 8 12 16 24 32; do
    n=$(( 65536 / a * a ))
    timeout 200 python bench.py --only-headline --agents $n --agents-per-env $a --steps 200 --warmup 20 > $OUT/many_$a.log 2>&1; line $OUT/many_$a.log "A=$a product"
  done; } | tee $OUT/many_agents.txt
  ;;
fan)
  X="env F110_LIB_VARIANT=experimental"
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  experimental build, bench.py $H --agents n, F110_EXP=integrate_fan=f"
  for n in 1024 2048 4096 8192 16384 32768 65536 131072; do for f in 0 1; do
    F110_EXP=integrate_fan=$f timeout 200 $X python bench.py $H --agents $n > $OUT/fan_n${n}_f$f.log 2>&1; line $OUT/fan_n${n}_f$f.log "agents $n integrate_fan $f"
  done; done; } | tee $OUT/fan_sweep.txt
  ;;
dropin)
  timeout 500 python tools/debug/dropin_rate.py 2048,32768 > $OUT/dropin_rate.txt 2>&1; cat $OUT/dropin_rate.txt
  ;;
gatherlegs)   # the gather legs at world size 1 (RCCL initialises, every leg's data checked): f64 / f32 / root, in stream / overlapped
  timeout 600 python bench.py --gather-legs --no-cpu-baseline --no-dropin --secondary 0 --no-config5 --fixed-pose-steps 0 --steady-steps 0 --steps 50 --warmup 10 > $OUT/bench_gather_legs.log 2>&1
  grep -h "^{" $OUT/bench_gather_legs.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); mg = d['multi_gpu']
    print('headline %.4f ms' % d['ms_per_step'], ' rccl_ranks', mg.get('rccl_ranks'), ' error', mg.get('gather_error'))
    for k in ('gather', 'gather_overlap', 'gather_f32', 'gather_f32_overlap', 'gather_root', 'gather_root_f32_overlap'):
        if k in mg: print('%-26s %.4f ms/step  ok %s' % (k, mg[k]['ms_per_step'], mg[k]['gather_ok']))
" | tee $OUT/gather_legs.txt
  ;;
bench)
  ( time timeout 900 python bench.py > $OUT/bench_default.log 2>$OUT/bench_default.err ) 2> $OUT/bench_default.time; echo "bench exit $?" >> $OUT/bench_default.log; tail -3 $OUT/bench_default.time
  tail -c 3000 $OUT/bench_default.log
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --secondary 0 --no-config5 --fixed-pose-steps 0 > $OUT/bench_driver_form.log 2>&1; line $OUT/bench_driver_form.log "driver form --steps 20 --warmup 5"
  ;;
prof)
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_stats -o stats -- python $R/bench.py $H > $OUT/prof_stats.log 2>&1
  python $R/tools/summarize_prof.py stats $OUT/prof_stats $OUT/kernel_stats.txt 300; rm -rf $OUT/prof_stats
  timeout 400 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_stats4k -o stats -- python $R/bench.py $H --agents 4096 --groups 1 > $OUT/prof_stats4k.log 2>&1
  python $R/tools/summarize_prof.py stats $OUT/prof_stats4k $OUT/kernel_stats_4096.txt 300; rm -rf $OUT/prof_stats4k
  timeout 400 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_stats5 -o stats -- python $R/bench.py $C5 > $OUT/prof_stats5.log 2>&1
  python $R/tools/summarize_prof.py stats $OUT/prof_stats5 $OUT/kernel_stats_cfg5.txt 100; rm -rf $OUT/prof_stats5
  timeout 300 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_vec -o stats -- python $R/tools/debug/vecenv_loop.py 2048 1000 > $OUT/prof_vec.log 2>&1
  python $R/tools/summarize_prof.py stats $OUT/prof_vec $OUT/kernel_stats_vecenv2048.txt 1000; rm -rf $OUT/prof_vec
  cd "$R"; head -14 $OUT/kernel_stats.txt; tail -8 $OUT/kernel_stats_4096.txt; tail -8 $OUT/kernel_stats_vecenv2048.txt
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
  for cfg in "4096:--agents 4096 --groups 1" "cfg5:--agents 65536 --beams 4096 --map-tiles 2 --steps 100 --warmup 20 --preroll 100"; do
    tagc=${cfg%%:*}; argsc=${cfg#*:}; n=300; [ "$tagc" = cfg5 ] && n=100
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --pmc $c --kernel-include-regex "k_scan_rays|k_scan_dirs" -T -f csv -d $OUT/tr_$c -o p -- python $R/bench.py --only-headline $argsc > $OUT/tr_${tagc}_$c.log 2>&1
      python $R/tools/summarize_prof.py pmc $OUT/tr_$c $OUT/traffic_${tagc}_$c.json - $n
      rm -rf $OUT/tr_$c
    done
    timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-include-regex "k_scan_rays|k_scan_dirs" -T -f csv -d $OUT/tr_vm -o p -- python $R/bench.py --only-headline $argsc > $OUT/tr_${tagc}_vm.log 2>&1
    python $R/tools/summarize_prof.py pmc $OUT/tr_vm $OUT/traffic_${tagc}_VMEM.json - $n
    rm -rf $OUT/tr_vm
  done
  cd "$R"; ls $OUT/pmc_pass*.json $OUT/traffic_*.json 2>/dev/null | wc -l
  ;;
pmc4096)
  cd /tmp
  i=0
  for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES" "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_TA_BUSY GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $ctrs --kernel-include-regex "k_scan_rays" -T -f csv -d $OUT/p4k_$i -o p -- python $R/bench.py $H --agents 4096 --groups 1 > $OUT/p4k_$i.log 2>&1
    python $R/tools/summarize_prof.py pmc $OUT/p4k_$i $OUT/pmc_4096_pass$i.json - 300
    rm -rf $OUT/p4k_$i
  done
  cd "$R"; python - <<'PYEOF'
import json, glob
m = {}
for f in sorted(glob.glob("gpurun_out/pmc_4096_pass*.json")):
    for k, r in json.load(open(f)).items():
        m.update(r["mean_per_dispatch"]); meta = r["meta"]; csrc = r.get("csrc")
cyc = m["GRBM_GUI_ACTIVE"] / 8.0
tasks = 4096 * 17
print("k_scan_rays_agent at 4096 agents: kernel cycles %.0f (%.1f us at 2.4 GHz)  VGPR %s SGPR %s" % (cyc, cyc / 2.4e3, meta.get("VGPR_Count"), meta.get("SGPR_Count")))
print("per task: VALU %.1f  SALU %.1f  VMEM rd %.2f wr %.2f   waves %d" % (m["SQ_INSTS_VALU"] / tasks, m["SQ_INSTS_SALU"] / tasks, m["SQ_INSTS_VMEM_RD"] / tasks, m["SQ_INSTS_VMEM_WR"] / tasks, m["SQ_WAVES"]))
print("TA busy %.3f  TD busy %.3f  TCP hit %.3f  TCC hit %.3f  wave cycles per wave %.0f" % (m["TA_TA_BUSY_sum"] / 256.0 / cyc, m["TD_TD_BUSY_sum"] / 256.0 / cyc, 1 - m["TCP_TCC_READ_REQ_sum"] / m["TCP_TOTAL_CACHE_ACCESSES_sum"], m["TCC_HIT_sum"] / m["TCC_REQ_sum"], 4 * m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"]))
m["csrc"] = csrc
m["what"] = "k_scan_rays_agent, BASELINE configs[1] (4096 agents): PMC means over the 300 timed dispatches (tools/gpu_r4.sh pmc4096)"
json.dump(m, open("gpurun_out/pmc_4096.json", "w"), indent=1, sort_keys=True)
PYEOF
  ;;
pmccfg5)
  cd /tmp
  i=0
  for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS" "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_TA_BUSY GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $ctrs --kernel-include-regex "k_scan_dirs" -T -f csv -d $OUT/pc5_$i -o p -- python $R/bench.py $C5 > $OUT/pc5_$i.log 2>&1
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
tabench)
  hipcc --offload-arch=gfx950 -O3 -o /tmp/ta_bench tools/debug/ta_bench.hip > $OUT/ta_bench_build.log 2>&1
  { echo "# $(date -u) tools/debug/ta_bench.hip on this box"; cat $OUT/box.txt; timeout 200 /tmp/ta_bench; } > $OUT/ta_bench.txt 2>&1; tail -20 $OUT/ta_bench.txt
  ;;
esac
done
