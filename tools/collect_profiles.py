#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of the last GPU session (gpurun_out/, scratch) into profiles/
(tracked) under a round tag, and refresh profiles/pmc_scan.json — the HBM-traffic record that
bench.py reports as roofline.traffic.

    python tools/collect_profiles.py r01 [agents beams layout]

HBM bytes per k_scan_rays launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in
KiB and, on gfx950, FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md §HBM), hence x2.
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out")
DST = os.path.join(ROOT, "profiles")


def main():
    tag = sys.argv[1]
    agents, beams, layout = (int(v) for v in (sys.argv[2:5] if len(sys.argv) >= 5 else (65536, 1080, 0)))
    os.makedirs(DST, exist_ok=True)
    shutil.copyfile(os.path.join(SRC, "kernel_stats.txt"), os.path.join(DST, "%s_kernel_stats.txt" % tag))
    merged = {}
    for i in range(1, 12):
        p = os.path.join(SRC, "pmc_pass%d.json" % i)
        if not os.path.isfile(p):
            continue
        for kern, rec in json.load(open(p)).items():
            m = merged.setdefault(kern, {"dispatches": rec["dispatches"], "mean_per_dispatch": {}, "meta": rec["meta"]})
            m["mean_per_dispatch"].update(rec["mean_per_dispatch"])
    json.dump(merged, open(os.path.join(DST, "%s_pmc.json" % tag), "w"), indent=1, sort_keys=True)
    for name in ("bench_default.log", "box.txt"):
        if os.path.isfile(os.path.join(SRC, name)):
            shutil.copyfile(os.path.join(SRC, name), os.path.join(DST, "%s_%s" % (tag, name.replace(".log", ".json") if name.endswith(".log") else name)))
    scan = (merged.get("k_scan_rays_agent") or merged.get("k_scan_rays") or {}).get("mean_per_dispatch", {})
    if "FETCH_SIZE" in scan and "WRITE_SIZE" in scan:
        rec_path = os.path.join(DST, "pmc_scan.json")
        rec = json.load(open(rec_path)) if os.path.isfile(rec_path) else {}
        rec["agents=%d,beams=%d,layout=%d" % (agents, beams, layout)] = {
            "round": tag, "FETCH_SIZE_KiB": scan["FETCH_SIZE"], "WRITE_SIZE_KiB": scan["WRITE_SIZE"],
            "hbm_bytes_per_launch": (2.0 * scan["FETCH_SIZE"] + scan["WRITE_SIZE"]) * 1024.0,
            "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request)"}
        json.dump(rec, open(rec_path, "w"), indent=1, sort_keys=True)
    print("profiles/%s_* written" % tag)


if __name__ == "__main__":
    main()
