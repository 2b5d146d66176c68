#!/bin/bash
# PMC passes on the scan kernel only:  gpurun -- 'bash tools/pmc.sh "<bench args>" "CTR CTR ..." "CTR ..."'
cd "${GRAFT_REPO_ROOT:-.}"; R="$PWD"; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
BARGS="$1"; shift
cd /tmp; i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --kernel-include-regex k_scan_rays -T -f csv -d $OUT/pmc_$i -o p -- python $R/bench.py --no-cpu-baseline --secondary 0 --fixed-pose-steps 0 --no-profile-events --steps 12 --warmup 2 $BARGS > $OUT/pmc_$i.log 2>&1
  python $R/tools/summarize_prof.py pmc $OUT/pmc_$i $OUT/pmc_pass$i.json k_scan_rays
  rm -rf $OUT/pmc_$i
  python - $OUT/pmc_pass$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    print(k[:30], v['dispatches'], {a:round(b,1) for a,b in v['mean_per_dispatch'].items()})
PY
done
