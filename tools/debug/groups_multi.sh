cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1 || exit 1
{ python -c "from f1tenth_gym_amd import build; print('# csrc', build.src_hash())"
for A in 16 8 4; do for G in 1 2 3 4; do
 N=65536; [ $A = 12 ] && N=65520
 F110_LIB_VARIANT=experimental timeout 120 python bench.py --only-headline --agents $N --agents-per-env $A --groups $G --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('A=$A groups=$G  %.2f M/s  %.4f ms' % (d['value']/1e6, d['ms_per_step']))
"; done; done; } | tee gpurun_out/groups_multi.txt
