#!/bin/bash
# HISTORICAL (round 2).  The F110_* environment switches used below were read by the library in round 2; since
# round 3 the product library reads no environment and the A/B switches live in the experimental build behind
# f110_exp_set (F110_LIB_VARIANT=experimental F110_EXP="key=value,...", see tools/gpu_r3.sh for the current sweeps).
# Round-2 GPU sessions (one gpurun call each; GPU minutes are scarce, so every phase is bounded):
#   gpurun --timeout 1500 -- 'bash tools/gpu_r2.sh test ab'
#   gpurun --timeout 1500 -- 'bash tools/gpu_r2.sh bench prof pmc'
cd "${GRAFT_REPO_ROOT:-.}"
R="$PWD"; export TMPDIR=/tmp
OUT=$R/gpurun_out; mkdir -p $OUT
{ rocminfo | grep -E "Marketing Name|gfx9|Compute Unit" | head -8; nproc; lscpu | grep "Model name" | head -1; } > $OUT/box.txt 2>&1
python -c "import __graft_entry__ as g; print(g.build())" > $OUT/build.log 2>&1
H="--only-headline --steps 300 --warmup 30"
line() { grep -h '^{' "$1" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline', {})
    print('%-28s %8.2f M/s  %.4f ms/step  resets %d' % ('$2', d['value']/1e6, d['ms_per_step'], d['config']['env_resets_in_timed_region']))
"; }
for MODE in "$@"; do
case $MODE in
test)
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -x --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
  tail -25 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log
  ;;
ta)
  timeout 300 tools/debug/ta_bench > $OUT/ta_bench.txt 2>&1; tail -24 $OUT/ta_bench.txt
  ;;
ab)
  # env groups x batch size, collide placement (one JSON line each, headline leg only)
  for n in 4096 16384 65536; do for g in 1 2 4; do
    timeout 200 python bench.py $H --agents $n --groups $g > $OUT/ab_n${n}_g${g}.log 2>&1; line $OUT/ab_n${n}_g${g}.log "agents $n groups $g"
  done; done
  timeout 200 python bench.py $H --agents 4096 --groups 8 > $OUT/ab_n4096_g8.log 2>&1; line $OUT/ab_n4096_g8.log "agents 4096 groups 8"
  F110_COLLIDE_MODE=2 timeout 200 python bench.py $H --agents 65536 --groups 1 > $OUT/ab_n65536_inline.log 2>&1; line $OUT/ab_n65536_inline.log "65536 g1 collide inline"
  F110_COLLIDE_MODE=1 timeout 200 python bench.py $H --agents 65536 --groups 1 > $OUT/ab_n65536_fused.log 2>&1; line $OUT/ab_n65536_fused.log "65536 g1 collide fused"
  F110_COLLIDE_MODE=1 timeout 200 python bench.py $H --agents 4096 --groups 1 > $OUT/ab_n4096_fused.log 2>&1; line $OUT/ab_n4096_fused.log "4096 g1 collide fused"
  timeout 200 python bench.py $H --agents 65536 --noise table > $OUT/ab_n65536_table.log 2>&1; line $OUT/ab_n65536_table.log "65536 noise table"
  ;;
graph)
  for n in 4096 16384 65536; do for g in 0 1; do
    timeout 200 python bench.py $H --agents $n --graph $g > $OUT/graph_n${n}_g${g}.log 2>&1; line $OUT/graph_n${n}_g${g}.log "agents $n graph $g"
  done; done
  ;;
gaps)
  cd /tmp
  for n in 4096 65536; do
    timeout 300 rocprofv3 --kernel-trace -T -f csv -d $OUT/trace_$n -o t -- python $R/bench.py --only-headline --steps 100 --warmup 10 --agents $n > $OUT/trace_$n.log 2>&1
    python $R/tools/summarize_prof.py gaps $OUT/trace_$n $OUT/gaps_$n.txt; rm -rf $OUT/trace_$n; echo "agents $n"; cat $OUT/gaps_$n.txt
  done
  cd "$R"
  ;;
pairfin)
  for n in 1024 4096 16384 65536; do for c in 0 3; do
    F110_COLLIDE_MODE=$c timeout 200 python bench.py $H --agents $n > $OUT/pf_n${n}_c${c}.log 2>&1; line $OUT/pf_n${n}_c${c}.log "agents $n collide-mode $c"
  done; done
  ;;
order)
  for n in 1024 2048 4096 8192; do for o in 0 1; do
    F110_TASK_ORDER=$o timeout 200 python bench.py $H --agents $n > $OUT/order_n${n}_o${o}.log 2>&1; line $OUT/order_n${n}_o${o}.log "agents $n order $o"
  done; done
  for thr in 24 32 64 96; do F110_TASK_ORDER=1 F110_TASK_THR=$thr timeout 200 python bench.py $H --agents 4096 > $OUT/order_thr$thr.log 2>&1; line $OUT/order_thr$thr.log "4096 thr $thr"; done
  ;;
win)
  for n in 4096 16384 65536; do for l in 3 4; do
    timeout 200 python bench.py $H --agents $n --layout $l > $OUT/win_n${n}_l${l}.log 2>&1; line $OUT/win_n${n}_l${l}.log "agents $n layout $l"
  done; done
  ;;
bench)
  timeout 900 python bench.py > $OUT/bench_default.log 2>&1; echo "bench exit $?" >> $OUT/bench_default.log
  tail -c 6000 $OUT/bench_default.log
  ;;
prof)
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_stats -o stats -- python $R/bench.py $H > $OUT/prof_stats.log 2>&1
  python $R/tools/summarize_prof.py stats $OUT/prof_stats $OUT/kernel_stats.txt; rm -rf $OUT/prof_stats
  timeout 400 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_stats4k -o stats -- python $R/bench.py $H --agents 4096 > $OUT/prof_stats4k.log 2>&1
  python $R/tools/summarize_prof.py stats $OUT/prof_stats4k $OUT/kernel_stats_4096.txt; rm -rf $OUT/prof_stats4k
  cd "$R"; cat $OUT/kernel_stats.txt | head -30
  ;;
pmc)
  cd /tmp
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_TA_BUSY GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $ctrs --kernel-include-regex "k_scan_rays|k_finalize|k_integrate|k_collide" -T -f csv -d $OUT/pmc_$i -o p -- python $R/bench.py $H > $OUT/pmc_$i.log 2>&1
    python $R/tools/summarize_prof.py pmc $OUT/pmc_$i $OUT/pmc_pass$i.json
    rm -rf $OUT/pmc_$i
  done
  cd "$R"
  ;;
esac
done
