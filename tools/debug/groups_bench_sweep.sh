#!/bin/bash
# bench.py's steady-regime workload, one block against two env blocks, over agents per env and batch size
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1 || exit 1
{ echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')  bench.py --only-headline --agents N --agents-per-env A --groups G --steps 300 --warmup 30: ms per step"
for A in 1 2 4 8; do for N in 1024 2048 3072 4096 6144 8192 16384 32768 65536; do
  r=""
  for G in 1 2; do
    v=$(timeout 120 python bench.py --only-headline --agents $N --agents-per-env $A --groups $G --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('%.4f' % json.loads(l)['ms_per_step'])
")
    r="$r $v"
  done
  echo "A $A N $N  one block / two blocks: $r" | awk '{printf "%s  %+.1f %%\n", $0, ($(NF-1)/$NF-1)*100}'
done; done; } | tee gpurun_out/groups_bench_sweep.txt
