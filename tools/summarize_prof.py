#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats / counter collection) into small text+JSON
summaries that are cheap to pull from the GPU box and to commit under profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def kernel_stats(d, out, last_n=0):
    """rocprofv3 --kernel-trace --stats: the tool's own per-kernel table (every dispatch of the process) and, with
    last_n > 0, the same statistics over the LAST last_n dispatches of every kernel from the trace — the bench's
    timed steps, without its pre-roll and warm-up launches (what roofline.kernel_ms_avg is measured over)."""
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    lines = ["# csrc %s" % src_hash()]
    for f in files:
        rows = list(csv.DictReader(open(f)))
        lines.append("# %s — every dispatch of the process" % os.path.basename(f))
        lines.append("%-70s %8s %12s %12s %12s %12s %7s" % ("Name", "Calls", "Total(ns)", "Avg(ns)", "Min(ns)", "Max(ns)", "Pct"))
        for r in rows:
            lines.append("%-70s %8s %12s %12.1f %12s %12s %7s" % (r["Name"][:70], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]),
                                                                r["MinNs"], r["MaxNs"], r["Percentage"]))
    if last_n > 0:
        per = defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                per[r["Kernel_Name"].split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        lines.append("# the last %d dispatches of every kernel that has that many (the timed steps), from the kernel trace" % last_n)
        lines.append("%-70s %8s %12s %12s %12s %12s" % ("Name", "Calls", "Total(ns)", "Avg(ns)", "Min(ns)", "Max(ns)"))
        tot = {}
        for k, v in per.items():
            if len(v) >= last_n:
                dur = [x[1] for x in sorted(v)[-last_n:]]
                tot[k] = dur
        for k, dur in sorted(tot.items(), key=lambda kv: -sum(kv[1])):
            lines.append("%-70s %8d %12d %12.1f %12d %12d" % (k[:70], len(dur), sum(dur), sum(dur) / len(dur), min(dur), max(dur)))
    open(out, "w").write("\n".join(lines) + "\n")


def src_hash():
    """identity of the kernel sources the profiled library was built from (f1tenth_gym_amd/build.py)"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from f1tenth_gym_amd import build
    return build.src_hash()


def counters(d, out, kernel_filter=None, last_n=0):
    """mean counter values per dispatch and kernel.  last_n > 0: only the last `last_n` dispatches of every
    kernel — the bench's TIMED steps, without the pre-roll and warm-up dispatches of the same process."""
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = []
    for f in files:
        rows.extend(csv.DictReader(open(f)))
    keep = None
    if last_n > 0:
        ids = defaultdict(set)
        for r in rows:
            ids[r["Kernel_Name"].split("(")[0][:60]].add(int(r["Dispatch_Id"]))
        keep = {k: set(sorted(v)[-last_n:]) for k, v in ids.items()}
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    meta = {}
    for r in rows:
        k = r["Kernel_Name"].split("(")[0][:60]
        if kernel_filter and kernel_filter not in k:
            continue
        if keep is not None and int(r["Dispatch_Id"]) not in keep[k]:
            continue
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
        meta[k] = {x: r.get(x) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    res = {k: {"dispatches": max(v[1] for v in c.values()), "mean_per_dispatch": {n: v[0] / v[1] for n, v in c.items()}, "meta": meta[k],
               "csrc": src_hash(), "window": ("last %d dispatches (the timed steps)" % last_n) if last_n > 0 else "every dispatch of the process"}
           for k, c in acc.items()}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


def gaps(d, out):
    """idle time between consecutive kernels of the step chain (integrate -> scan -> finalize -> integrate)
    from a --kernel-trace CSV: where the step's time goes besides the kernels themselves"""
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
    rows.sort()
    chain = [r for r in rows if r[2].startswith(("k_integrate", "k_scan_rays", "k_finalize", "k_expand", "k_reset_collided"))]
    acc = defaultdict(list)
    for a, b in zip(chain[:-1], chain[1:]):
        acc["%s -> %s" % (a[2][:18], b[2][:18])].append(b[0] - a[1])
    lines = ["%-44s %8s %10s %10s" % ("gap", "count", "mean(us)", "median(us)")]
    for k, v in sorted(acc.items()):
        v = sorted(v)
        lines.append("%-44s %8d %10.2f %10.2f" % (k, len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3))
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    mode, d, out = sys.argv[1:4]
    if mode == "stats":   # stats <dir> <out> [last N dispatches]
        kernel_stats(d, out, int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    elif mode == "gaps":
        gaps(d, out)
    else:   # pmc <dir> <out> [kernel-name filter | -] [last N dispatches]
        kf = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] != "-" else None
        counters(d, out, kf, int(sys.argv[5]) if len(sys.argv) > 5 else 0)
