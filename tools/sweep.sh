#!/bin/bash
# quick perf sweep of scan launch geometry:  gpurun -- 'bash tools/sweep.sh'
cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
SHORT="--no-cpu-baseline --secondary 0 --fixed-pose-steps 0 --steps 150 --warmup 15"
for cfg in "$@"; do
  set -- $cfg
  echo "== $cfg"
  timeout 200 python bench.py $SHORT $(echo $cfg | tr ',' ' ') 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d.get('roofline', {})
    print('%.2fM/s ms/step %.3f scan %.3f int+col %.3f fin %.3f' % (d['value']/1e6, d['ms_per_step'], r.get('kernel_ms_avg', 0), r.get('integrate_collide_ms_avg', 0), r.get('finalize_ms_avg', 0)))
"
done 2>&1 | tee $OUT/sweep.log
