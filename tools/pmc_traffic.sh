#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE: two passes) of the scan kernel for one bench configuration:
#   gpurun -- 'bash tools/pmc_traffic.sh <tag> <bench args...>'   ->  gpurun_out/traffic_<tag>.json
cd "${GRAFT_REPO_ROOT:-.}"; R="$PWD"; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
TAG=$1; shift
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-include-regex "k_scan_rays|k_scan_dirs" -T -f csv -d $OUT/tr_$c -o p -- python $R/bench.py --only-headline "$@" > $OUT/tr_$c.log 2>&1
  python $R/tools/summarize_prof.py pmc $OUT/tr_$c $OUT/traffic_${TAG}_$c.json
  rm -rf $OUT/tr_$c
done
python - $OUT $TAG <<'PY'
import json, sys
out, tag = sys.argv[1:3]
rec = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = json.load(open("%s/traffic_%s_%s.json" % (out, tag, c)))
    for k, v in d.items():
        rec.setdefault(k, {}).update(v["mean_per_dispatch"]); rec[k]["dispatches"] = v["dispatches"]
print(json.dumps(rec))
json.dump(rec, open("%s/traffic_%s.json" % (out, tag), "w"), indent=1)
PY
