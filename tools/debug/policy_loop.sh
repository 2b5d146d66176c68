#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1 || exit 1
echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  bench.py --only-headline --policy pure_pursuit --agents N --groups G: ms per step, one block / automatic"
for n in 4096 8192 16384 32768 65536; do r=""; for G in 1 0; do
  v=$(timeout 120 python bench.py --only-headline --policy pure_pursuit --agents $n --groups $G --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): d = json.loads(l); print('%.4f (%s)' % (d['ms_per_step'], d['config'].get('env_blocks_per_step')))
"); r="$r $v"; done; echo "agents $n: $r"; done
