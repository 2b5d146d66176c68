#!/bin/bash
# After a `tools/gpu_r6.sh test prof pmc tabench manyagents rates cfg3 tinyab latency envprof [soak fuzzlong] [bench]` session: the summaries
# of gpurun_out/ (scratch) into profiles/ (tracked), every file with the source hash it was measured on in its first lines.
cd "$(dirname "$0")/.."
python tools/collect_profiles.py r06 65536 1080 3 | tail -2
for f in kernel_stats_cfg5.txt:r06_kernel_stats_cfg5.txt ta_bench.txt:r06_ta_bench.txt box.txt:r06_box.txt many_agents.txt:r06_many_agents.txt rates.txt:r06_rates.txt \
         cfg3_one_device.txt:r06_cfg3_one_device.txt envprof.txt:r06_f110env_final.txt tiny_tail_ab.txt:r06_tiny_tail_ab.txt soak.txt:r06_soak.txt; do
  src=gpurun_out/${f%%:*}; dst=profiles/${f##*:}
  [ -f "$src" ] && cp "$src" "$dst"
done
python - <<'PY'
import os, re
csrc = __import__("f1tenth_gym_amd.build", fromlist=["x"]).src_hash()
# launch latency: the session's body, the Python-cost appendix kept
old = open("profiles/r06_launch_latency.txt").read()
if os.path.isfile("gpurun_out/launch_latency.txt"):
    new = open("gpurun_out/launch_latency.txt").read()
    i = old.find("## tools/debug/f110env_python_cost.py")
    open("profiles/r06_launch_latency.txt", "w").write(new.rstrip("\n") + "\n" + (old[i:] if i >= 0 else ""))
# tiny A/B: history kept, one section per source hash
p = "profiles/r06_tiny_ab.txt"
s = open(p).read()
if os.path.isfile("gpurun_out/tiny_ab.txt"):
    new = open("gpurun_out/tiny_ab.txt").read()
    if csrc in new and csrc not in s:
        s += "\n## the same A/B on the final sources\n" + new
        open(p, "w").write(s)
# fuzz: the long run of this session first, the earlier hashes' runs kept below
p = "profiles/r06_fuzz.txt"
if os.path.isfile("gpurun_out/fuzz_long.txt"):
    new = open("gpurun_out/fuzz_long.txt").read()
    s = open(p).read()
    if csrc in new and csrc not in s:
        open(p, "w").write(new.rstrip("\n") + "\n## the same and a second set of seed ranges on earlier sources of this round (the kernels differ from the final ones as git shows)\n" + s)
PY
git status --short profiles | head -30
