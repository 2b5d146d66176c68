#!/bin/bash
# after `gpurun -- 'bash tools/gpu_r4.sh prof pmc pmccfg5 pmc4096 tabench manyagents dropin bench'`: everything profiles/r04_* holds,
# (r04_many_agents.txt carries hand-written notes below its table: merge by hand)
# from gpurun_out/ (tools/collect_profiles.py writes the merged PMC / issue-floor files, the rest is copied as it is).
# bench.py reads profiles/pmc_scan.json and r04_issue_floor.json (same source hash only), so run `gpu_r4.sh bench` once more
# AFTER this script and call it again: the committed bench line then carries `traffic` and `issue_floor_frac`.
cd "$(dirname "$0")/.."
python tools/collect_profiles.py r04 65536 1080 3
for f in kernel_stats_cfg5.txt kernel_stats_vecenv2048.txt pmc_4096.json pmc_cfg5.json ta_bench.txt dropin_rate.txt box.txt; do
  [ -f gpurun_out/$f ] && cp gpurun_out/$f profiles/r04_$f
done
rm -f profiles/r04_late_*.txt profiles/r04_latency_sizes.txt profiles/r04_scan_timeline_4096.txt profiles/r04_soak_fuzz.txt profiles/r04_vecenv_rate.txt
grep -l "csrc" profiles/r04_* | while read f; do printf "%-48s %s\n" "$f" "$(grep -o 'csrc[": =]*[0-9a-f]\{16\}' "$f" | grep -o '[0-9a-f]\{16\}' | sort -u | tr '\n' ' ')"; done
