#!/bin/bash
# PMC passes over k_finalize_multi (A cars per env, one block per step): where do its cycles go?
# usage: pmc_multi.sh A
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; OUT=$R/gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1 || exit 1
A=${1:-16}; N=$(( 65536 / A * A ))
cd /tmp
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --kernel-include-regex "k_finalize_multi" -T -f csv -d $OUT/pm_$i -o p -- python $R/bench.py --only-headline --agents $N --agents-per-env $A --groups 1 --steps 100 --warmup 10 > $OUT/pm_$i.log 2>&1
  python $R/tools/summarize_prof.py pmc $OUT/pm_$i $OUT/pmc_multi_pass$i.json - 100
  rm -rf $OUT/pm_$i
done
cd $R; python - "$A" <<'PY'
import json, glob, sys
m = {}
for f in sorted(glob.glob("gpurun_out/pmc_multi_pass*.json")):
    for k, r in json.load(open(f)).items():
        m.update(r["mean_per_dispatch"]); meta = r["meta"]
cyc = m["GRBM_GUI_ACTIVE"] / 8.0 if "GRBM_GUI_ACTIVE" in m else None
w = m["SQ_WAVES"]
print("k_finalize_multi A=%s: waves %d  VGPR %s LDS %s  kernel cycles %s" % (sys.argv[1], w, meta.get("VGPR_Count"), meta.get("LDS_Block_Size"), cyc))
for k in sorted(m): print("  %-28s %14.1f   per wave %10.1f" % (k, m[k], m[k] / w))
PY
