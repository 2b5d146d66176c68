cd /tmp; export TMPDIR=/tmp
for a in 1 3 4; do
  python $GRAFT_REPO_ROOT/bench.py --only-headline --agents 65520 --agents-per-env $a --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('A=$a', round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), 'ms')
"
done
timeout 300 rocprofv3 --kernel-trace --stats -T -f csv -d /tmp/prof_a4 -o stats -- python $GRAFT_REPO_ROOT/bench.py --only-headline --agents 65536 --agents-per-env 4 --steps 200 --warmup 20 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/summarize_prof.py stats /tmp/prof_a4 /tmp/a4_stats.txt 200; sed -n '/last 200/,$p' /tmp/a4_stats.txt | head -8
