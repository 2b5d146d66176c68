#!/bin/bash
# two env blocks of unequal size (experimental build, F110_EXP=group_split=<percent of the envs in the first block>)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1 || exit 1
echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  experimental build, bench.py --only-headline --groups 2, ms per step at group_split 50 / 55 / 60 / 67"
for cfg in "65536 16" "65536 8" "65536 4" "65536 1" "16384 2" "32768 2" "65536 2"; do set -- $cfg
  n=$(( $1 / $2 * $2 )); r=""
  for S in 0 55 60 67; do
    v=$(F110_LIB_VARIANT=experimental F110_EXP="group_split=$S" timeout 120 python bench.py --only-headline --agents $n --agents-per-env $2 --groups 2 --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('%.4f' % json.loads(l)['ms_per_step'])
"); r="$r $v"; done
  echo "agents $n A $2: $r"
done
