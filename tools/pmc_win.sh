#!/bin/bash
# PMC passes on the scan kernel of one layout:  gpurun -- 'bash tools/pmc_win.sh <layout> [extra bench args]'
cd "${GRAFT_REPO_ROOT:-.}"; R="$PWD"; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
L=$1; shift
cd /tmp; i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_TA_BUSY GRBM_GUI_ACTIVE" "MeanOccupancyPerCU VALUBusy MemUnitStalled"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --kernel-include-regex "k_scan_rays" -T -f csv -d $OUT/pmcw_$i -o p -- python $R/bench.py --only-headline --steps 60 --warmup 30 --layout $L "$@" > $OUT/pmcw_$i.log 2>&1
  python $R/tools/summarize_prof.py pmc $OUT/pmcw_$i $OUT/pmcw_l${L}_pass$i.json k_scan_rays
  rm -rf $OUT/pmcw_$i
  python - $OUT/pmcw_l${L}_pass$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    print(k[:30], v['dispatches'], {a:round(b,1) for a,b in v['mean_per_dispatch'].items()}, v['meta'])
PY
done
