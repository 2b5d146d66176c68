"""From a rocprofv3 --kernel-trace --hip-trace run (csv) of tools/debug/f110env_loop.py: for every k_step_tiny dispatch, the time from the
host's hipLaunchKernel call (begin / end) to the kernel's first wave and the kernel's own span — the timestamps share one clock domain.
    python tools/debug/launch_gap.py <dir with *_kernel_trace.csv and *_hip_api_trace.csv>"""
import csv, glob, os, sys
import numpy as np
d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
ht = glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)[0]
kern = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], int(r.get("Correlation_Id", 0) or 0)) for r in csv.DictReader(open(kt))]
api = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], int(r.get("Correlation_Id", 0) or 0)) for r in csv.DictReader(open(ht))]
launches = {c: (s, e) for s, e, f, c in api if "LaunchKernel" in f}
rows = []
for s, e, name, c in kern:
    if "k_step_tiny" in name and c in launches:
        ls, le = launches[c]
        rows.append((s - ls, s - le, e - s))
rows = np.array(rows[len(rows) // 4:], dtype=np.float64) / 1e3
print("%d k_step_tiny dispatches (the last three quarters): microseconds, mean / p10 / p90" % len(rows))
for i, nme in enumerate(("hipLaunchKernel called -> first wave", "hipLaunchKernel returned -> first wave", "kernel begin -> end")):
    print("  %-42s %6.1f %6.1f %6.1f" % (nme, rows[:, i].mean(), np.percentile(rows[:, i], 10), np.percentile(rows[:, i], 90)))
# what else the host called per step
from collections import Counter
cnt = Counter(f for _, _, f, _ in api)
n = max(1, sum(1 for k in kern if "k_step_tiny" in k[2]))
print("HIP API calls per step:", {f: round(c / n, 2) for f, c in cnt.most_common(12)})
dur = {}
for s, e, f, _ in api:
    dur.setdefault(f, []).append(e - s)
print("mean duration (us):", {f: round(np.mean(v) / 1e3, 2) for f, v in dur.items() if len(v) > n // 2})
