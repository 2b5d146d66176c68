#!/bin/bash
# Round-5 GPU sessions (one gpurun call each; EVERY command under its own timeout — a hung rocprofv3 cost 20 GPU-minutes once):
#   gpurun --timeout 1500 -- 'bash tools/gpu_r5.sh test stream'
# modes: test testfast bench prof pmc tabench rates manyagents | lab measurements: stream spec finwave tiled pmcstream
#   the evidence of a source hash = 'test prof pmc tabench stream spec finwave tiled manyagents rates pmcstream', then
#   python tools/collect_profiles.py r05 65536 1080 3 (+ cp of the per-mode txt files), commit, then 'bench' (its line quotes the committed PMC traffic)
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
pmcstream)   # why the lane-refill scan loses: the same PMC passes over k_scan_rays_agent and k_scan_stream_agent (experimental build, 65 536 agents)
  cd /tmp
  for tag in "base:scan_stream=0" "stream:scan_stream=1,stream_block=64,stream_refill=48"; do
    nm=${tag%%:*}; ex=${tag#*:}; i=0
    for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS" "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
      i=$((i+1))
      F110_LIB_VARIANT=experimental F110_EXP=$ex timeout 300 rocprofv3 --pmc $ctrs --kernel-include-regex "k_scan_rays|k_scan_stream" -T -f csv -d $OUT/ps_$i -o p -- python $R/bench.py $H > $OUT/ps_$i.log 2>&1
      timeout 60 python $R/tools/summarize_prof.py pmc $OUT/ps_$i $OUT/pmcstream_${nm}_$i.json - 300
      rm -rf $OUT/ps_$i
    done
  done
  cd "$R"; python - <<'PY'
import json, glob, os
out = {}
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "pmcstream_*_*.json"))):
    nm = os.path.basename(f).split("_")[1]
    for k, v in json.load(open(f)).items():
        rec = out.setdefault(nm, {}).setdefault(k, {"csrc": v.get("csrc"), "dispatches": v["dispatches"], "mean_per_dispatch": {}, "meta": v.get("meta")})
        rec["mean_per_dispatch"].update(v["mean_per_dispatch"])
json.dump(out, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "pmc_stream_vs_base.json"), "w"), indent=1, sort_keys=True)
for nm, ks in out.items():
    for k, v in ks.items():
        m = v["mean_per_dispatch"]
        cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
        print(nm, k, "us %.0f" % (cyc / 2400.0), "vmem_rd %.3g wr %.3g valu %.3g salu %.3g lds %.3g" % (m.get("SQ_INSTS_VMEM_RD", 0), m.get("SQ_INSTS_VMEM_WR", 0), m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_SALU", 0), m.get("SQ_INSTS_LDS", 0)),
              "TA %.2f TD %.2f" % (m.get("TA_TA_BUSY_sum", 0) / 256 / max(cyc, 1), m.get("TD_TD_BUSY_sum", 0) / 256 / max(cyc, 1)),
              "L1 acc %.3g -> L2 req %.3g (L2 hit %.3f)" % (m.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0), m.get("TCP_TCC_READ_REQ_sum", 0), m.get("TCC_HIT_sum", 0) / max(m.get("TCC_REQ_sum", 1), 1)))
PY
  ;;
tiled)   # the scan on a 4x4-tiled copy of the PADDED table (experimental build)
  { echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  tools/debug/tiled_ab.py (parity) + tools/debug/stream_probe.py N pad_tiled=0|1 (experimental build)"
    F110_LIB_VARIANT=experimental timeout 200 python tools/debug/tiled_ab.py 2>&1 | tail -6
    for n in 65536 16384 4096; do for pt in 0 1; do F110_LIB_VARIANT=experimental timeout 90 python tools/debug/stream_probe.py $n pad_tiled=$pt 2>&1 | grep agents; done; done; } | tee $OUT/tiled_table.txt
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
