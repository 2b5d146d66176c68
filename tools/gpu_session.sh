#!/bin/bash
# One GPU-box session: parity tests, smoke, bench (+variants), rocprofv3 stats and PMC passes.
# Usage (from the build container):  gpurun --timeout 1800 -- 'bash tools/gpu_session.sh [quick]'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
{ rocminfo | grep -E "Marketing Name|gfx9|Compute Unit" | head -8; nproc; lscpu | grep "Model name"; free -g | head -2; } > $OUT/box.txt 2>&1
python -c "import __graft_entry__ as g; print(g.build())" > $OUT/build.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.log 2>&1; echo "bench exit $?" >> $OUT/bench_default.log
if [ "$1" != "quick" ]; then
  for layout in 0 1; do for blk in 64 128 256; do
    timeout 200 python bench.py --layout $layout --scan-block $blk --steps 100 --warmup 10 --no-cpu-baseline --secondary 0 > $OUT/bench_l${layout}_b${blk}.log 2>&1
  done; done
  timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --secondary 0 --no-noise --no-reset > $OUT/bench_nonoise_noreset.log 2>&1
fi
cd /tmp
R="$OLDPWD"
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_stats -o stats -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --secondary 0 --no-profile-events > $R/$OUT/prof_stats.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_fetch -o fetch -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --secondary 0 --no-profile-events > $R/$OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_write -o write -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --secondary 0 --no-profile-events > $R/$OUT/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $R/$OUT/pmc_sq -o sq -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --secondary 0 --no-profile-events > $R/$OUT/pmc_sq.log 2>&1
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $R/$OUT/pmc_cache -o cache -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --secondary 0 --no-profile-events > $R/$OUT/pmc_cache.log 2>&1
cd "$R"
find $OUT -name "*.csv" -size +3M -delete
find $OUT -name "*.db" -delete
ls -laR $OUT | head -80 > $OUT/listing.txt
tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; tail -2 $OUT/bench_default.log
