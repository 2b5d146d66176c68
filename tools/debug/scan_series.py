"""durations of every dispatch of the step kernels from a rocprofv3 --kernel-trace CSV directory: the last N steps as
a series (one line per step: integrate / scan / finalize in us, gap to the previous kernel) + a histogram of the scan"""
import csv, glob, os, sys
from collections import defaultdict
d, last = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows.extend(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps, cur = [], {}
prev_end = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    key = "int" if name.startswith("k_integrate") else ("scan" if name.startswith("k_scan") else ("fin" if name.startswith("k_finalize") else None))
    if key is None:
        prev_end = e
        continue
    cur[key] = (e - s) / 1e3
    cur[key + "_gap"] = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = e
    if key == "fin":
        steps.append(cur); cur = {}
steps = steps[-last:]
print("# step  integrate  gap  scan  gap  finalize  gap   (us)")
for i, s in enumerate(steps):
    print("%4d  %6.1f %5.1f  %6.1f %5.1f  %6.1f %5.1f" % (i, s.get("int", 0), s.get("int_gap", 0), s.get("scan", 0), s.get("scan_gap", 0), s.get("fin", 0), s.get("fin_gap", 0)))
sc = sorted(s.get("scan", 0) for s in steps)
n = len(sc)
print("# scan us: min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f mean %.1f" % (sc[0], sc[n // 10], sc[n // 2], sc[9 * n // 10], sc[min(n - 1, 99 * n // 100)], sc[-1], sum(sc) / n))
