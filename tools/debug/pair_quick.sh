#!/bin/bash
# headline / small-batch rates + finalize kernel time (kernel trace) for A = 2
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1 || exit 1
echo "# csrc $(python -c 'from f1tenth_gym_amd import build; print(build.src_hash())')"
for n in 65536 4096; do
  cd /tmp; rm -rf /tmp/pp
  timeout 200 rocprofv3 --kernel-trace --stats -T -f csv -d /tmp/pp -o s -- python $R/bench.py --only-headline --agents $n --groups 1 --steps 300 --warmup 30 > /tmp/pp.log 2>&1
  python - "$n" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/pp/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_finalize_pair' in r['Name'] or 'k_integrate' in r['Name'] or 'k_scan_rays_agent' in r['Name']:
        print("N=%6s  %-26s avg %.2f us over %s calls" % (sys.argv[1], r['Name'][:26], float(r['AverageNs']) / 1e3, r['Calls']))
PY
  cd $R
done
for i in 1 2 3; do for n in 65536 4096; do
timeout 120 python bench.py --only-headline --agents $n --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('N=$n  %.2f M/s  %.4f ms' % (d['value']/1e6, d['ms_per_step']))
"; done; done
