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
            m = merged.setdefault(kern, {"dispatches": rec["dispatches"], "mean_per_dispatch": {}, "meta": rec["meta"],
                                         "csrc": rec.get("csrc"), "window": rec.get("window")})
            m["mean_per_dispatch"].update(rec["mean_per_dispatch"])
            if rec.get("csrc") != m["csrc"]:
                raise SystemExit("PMC passes of different source trees: %s vs %s" % (rec.get("csrc"), m["csrc"]))
    json.dump(merged, open(os.path.join(DST, "%s_pmc.json" % tag), "w"), indent=1, sort_keys=True)
    for name in ("bench_default.log", "box.txt"):
        if os.path.isfile(os.path.join(SRC, name)):
            shutil.copyfile(os.path.join(SRC, name), os.path.join(DST, "%s_%s" % (tag, name.replace(".log", ".json") if name.endswith(".log") else name)))
    scan_rec = merged.get("k_scan_rays_agent") or merged.get("k_scan_rays") or {}
    scan = scan_rec.get("mean_per_dispatch", {})
    csrc, window = scan_rec.get("csrc"), scan_rec.get("window")
    if "FETCH_SIZE" in scan and "WRITE_SIZE" in scan:
        rec_path = os.path.join(DST, "pmc_scan.json")
        rec = json.load(open(rec_path)) if os.path.isfile(rec_path) else {}
        rec["agents=%d,beams=%d,layout=%d" % (agents, beams, layout)] = {
            "round": tag, "csrc": csrc, "window": window, "FETCH_SIZE_KiB": scan["FETCH_SIZE"], "WRITE_SIZE_KiB": scan["WRITE_SIZE"],
            "hbm_bytes_per_launch": (2.0 * scan["FETCH_SIZE"] + scan["WRITE_SIZE"]) * 1024.0,
            "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request)"}
        json.dump(rec, open(rec_path, "w"), indent=1, sort_keys=True)
    if "SQ_INSTS_VMEM_RD" in scan and "GRBM_GUI_ACTIVE" in scan:
        # the binding roofline of the scan kernel (bench.py roofline.issue_floor): wave-level vector-memory
        # instructions x the cheapest a 64-lane gather can issue on a gfx950 CU (tools/debug/ta_bench.hip)
        fl_path = os.path.join(DST, "%s_issue_floor.json" % tag)
        fl = json.load(open(fl_path)) if os.path.isfile(fl_path) else {}
        vm = scan["SQ_INSTS_VMEM_RD"] + scan.get("SQ_INSTS_VMEM_WR", 0.0)
        cyc = scan["GRBM_GUI_ACTIVE"] / 8.0   # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        key = "agents=%d,beams=%d,layout=%d" % (agents, beams, layout)
        fl.update({"what": "gather-issue floor of the scan kernel: wave-level vector-memory instructions per launch (rocprofv3 --pmc "
                           "SQ_INSTS_VMEM_RD + SQ_INSTS_VMEM_WR over the bench's timed steps) x the cheapest a 64-lane non-contiguous "
                           "gather issues on a gfx950 CU (r02_ta_bench.txt: 19.4-19.7 cycles per wave-level u64 load at 1-4 distinct "
                           "lines; no width, line count or active-lane mask measured is cheaper than ~17.5)",
                   "gather_cycles_per_wave_instr": 19.5, "cus": 256, "clock_mhz": 2400, "round": tag, "csrc": csrc, "window": window})
        fl.setdefault("vmem_instr_per_launch", {})[key] = vm
        fl.setdefault("pmc", {})[key] = {"kernel_cycles": cyc, "TA_TA_BUSY_frac": scan.get("TA_TA_BUSY_sum", 0.0) / 256.0 / cyc,
                                         "TD_TD_BUSY_frac": scan.get("TD_TD_BUSY_sum", 0.0) / 256.0 / cyc,
                                         "vmem_instr_per_64ray_task": vm / (agents * ((beams + 63) // 64)),
                                         "valu_instr_per_task": scan.get("SQ_INSTS_VALU", 0.0) / (agents * ((beams + 63) // 64)),
                                         "TCP_hit_frac": 1.0 - scan.get("TCP_TCC_READ_REQ_sum", 0.0) / max(scan.get("TCP_TOTAL_CACHE_ACCESSES_sum", 1.0), 1.0),
                                         "TCC_hit_frac": scan.get("TCC_HIT_sum", 0.0) / max(scan.get("TCC_REQ_sum", 1.0), 1.0)}
        json.dump(fl, open(fl_path, "w"), indent=1, sort_keys=True)
    print("profiles/%s_* written" % tag)


if __name__ == "__main__":
    main()
