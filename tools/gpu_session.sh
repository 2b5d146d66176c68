#!/bin/bash
# One GPU-box session.  Usage (from the build container):
#   gpurun --timeout 1800 -- 'bash tools/gpu_session.sh test'     parity tests + smoke + default bench
#   gpurun --timeout 1800 -- 'bash tools/gpu_session.sh perf'     bench variants
#   gpurun --timeout 1800 -- 'bash tools/gpu_session.sh prof'     rocprofv3 kernel stats + PMC passes
# Modes may be combined: 'test perf prof'.
cd "${GRAFT_REPO_ROOT:-.}"
R="$PWD"
export TMPDIR=/tmp
OUT=$R/gpurun_out; mkdir -p $OUT
MODES="${*:-test}"
{ rocminfo | grep -E "Marketing Name|gfx9|Compute Unit" | head -8; nproc; lscpu | grep "Model name" | head -1; } > $OUT/box.txt 2>&1
python -c "import __graft_entry__ as g; print(g.build())" > $OUT/build.log 2>&1
SHORT="--no-cpu-baseline --secondary 0 --fixed-pose-steps 0"
for MODE in $MODES; do
case $MODE in
test)
  timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
  timeout 900 python bench.py > $OUT/bench_default.log 2>&1; echo "bench exit $?" >> $OUT/bench_default.log
  ;;
perf)
  for layout in 0 1; do for blk in 64 128; do
    timeout 200 python bench.py --layout $layout --scan-block $blk --steps 200 --warmup 20 $SHORT > $OUT/bench_l${layout}_b${blk}.log 2>&1
  done; done
  timeout 200 python bench.py --steps 200 --warmup 20 $SHORT --no-noise --no-reset > $OUT/bench_nonoise_noreset.log 2>&1
  timeout 200 python bench.py --steps 200 --warmup 20 $SHORT --agents 4096 > $OUT/bench_4096.log 2>&1
  timeout 200 python bench.py --steps 200 --warmup 20 $SHORT --agents 16384 > $OUT/bench_16384.log 2>&1
  timeout 300 python bench.py --steps 100 --warmup 10 $SHORT --agents 262144 > $OUT/bench_262144.log 2>&1
  timeout 300 python bench.py --steps 100 --warmup 10 $SHORT --beams 4096 --agents 16384 > $OUT/bench_4096beams.log 2>&1
  ;;
prof)
  cd /tmp
  PB="python $R/bench.py $SHORT --no-profile-events"
  timeout 400 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_stats -o stats -- $PB --steps 100 --warmup 10 > $OUT/prof_stats.log 2>&1
  python $R/tools/summarize_prof.py stats $OUT/prof_stats $OUT/kernel_stats.txt
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_TA_BUSY GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "MeanOccupancyPerCU VALUBusy MemUnitStalled"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $ctrs --kernel-include-regex "k_scan_rays|k_finalize|k_integrate|k_collide" -T -f csv -d $OUT/pmc_$i -o p -- $PB --steps 12 --warmup 2 > $OUT/pmc_$i.log 2>&1
    python $R/tools/summarize_prof.py pmc $OUT/pmc_$i $OUT/pmc_pass$i.json
    rm -rf $OUT/pmc_$i
  done
  cd "$R"
  rm -rf $OUT/prof_stats
  ;;
esac
done
cd "$R"
for f in $OUT/pytest_gpu.log $OUT/smoke.log; do [ -f $f ] && tail -2 $f; done
grep -h '^{' $OUT/bench_*.log 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline', {})
    print('%.2fM/s ms/step %.3f scan %.3f int+col %.3f fin %.3f | %s' % (d['value']/1e6, d['ms_per_step'], r.get('kernel_ms_avg', 0), r.get('integrate_collide_ms_avg', 0), r.get('finalize_ms_avg', 0), d['config']['workload'][:40] + ' ' + d['config']['map_layout']))
"
