#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp
for P in 0 1 2; do
  F110_EXTRA_HIPCC_FLAGS="-DF110_PROBE=$P" python -c "from f1tenth_gym_amd import build; build.build(force=True)" >/dev/null 2>&1 || { echo build failed; exit 1; }
  for n in 65536 4096; do
  cd /tmp; rm -rf /tmp/pp
  timeout 200 rocprofv3 --kernel-trace --stats -T -f csv -d /tmp/pp -o s -- python $R/bench.py --only-headline --agents $n --groups 1 --steps 100 --warmup 10 --preroll 100 > /tmp/pp.log 2>&1
  python - "$P" "$n" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/pp/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_finalize_pair' in r['Name'] or 'k_integrate' in r['Name']:
        print("probe %4s N=%6s  %-30s avg %.1f us over %s calls" % (sys.argv[1], sys.argv[2], r['Name'][:30], float(r['AverageNs']) / 1e3, r['Calls']))
PY
  cd $R
  done
done
