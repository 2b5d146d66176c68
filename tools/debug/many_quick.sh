#!/bin/bash
# bench.py --only-headline over agents per env (product library), optional --groups G as $1
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1 || exit 1
echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())') groups ${1:-0}"
for a in 3 4 8 12 16 24; do
  n=$(( 65536 / a * a ))
  timeout 200 python bench.py --only-headline --agents $n --agents-per-env $a --groups ${1:-0} --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('A=$a  %.2f M/s  %.4f ms' % (d['value']/1e6, d['ms_per_step']))
"
done
