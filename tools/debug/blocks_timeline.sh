#!/bin/bash
# kernel timeline of the two env blocks (rocprofv3 --kernel-trace): do the blocks run in phase or interleaved?
# usage: blocks_timeline.sh "<bench args>" tag
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1 || exit 1
cd /tmp; rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace -f csv -d /tmp/tl -o t -- python $R/bench.py --only-headline $1 --steps 60 --warmup 10 > /tmp/tl.log 2>&1
python - "$2" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].split('<')[0].replace('void ', ''), r.get('Queue_Id', '?'), r.get('Stream_Id', '?')) for r in rows]
ks.sort()
ks = [k for k in ks if k[2].startswith(('k_scan', 'k_finalize', 'k_integrate'))]
tail = ks[-36:]
t0 = tail[0][0]
out = ["# last kernels of the timed loop: start / end (us from the first shown), kernel, queue, stream"]
for s, e, n, q, st in tail:
    out.append("%9.1f %9.1f  %6.1f us  %-24s q %s s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n, q, st))
open('%s/gpurun_out/blocks_timeline_%s.txt' % (sys.argv[0] and __import__('os').environ.get('GRAFT_REPO_ROOT', '.'), sys.argv[1]), 'w').write("\n".join(out) + "\n")
print("\n".join(out))
PY
