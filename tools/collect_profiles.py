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
                           "gather issues on a gfx950 CU (%s_ta_bench.txt, tools/debug/ta_bench.hip on the same box: 19.0-19.9 cycles per wave-level u64 load at 1-8 distinct "
                           "lines; no width, line count or active-lane mask measured is cheaper than ~17.4)" % tag,
                   "gather_cycles_per_wave_instr": 19.5, "cus": 256, "clock_mhz": 2400, "round": tag, "csrc": csrc, "window": window})
        fl.setdefault("vmem_instr_per_launch", {})[key] = vm
        fl.setdefault("pmc", {})[key] = {"kernel_cycles": cyc, "TA_TA_BUSY_frac": scan.get("TA_TA_BUSY_sum", 0.0) / 256.0 / cyc,
                                         "TD_TD_BUSY_frac": scan.get("TD_TD_BUSY_sum", 0.0) / 256.0 / cyc,
                                         "vmem_instr_per_64ray_task": vm / (agents * ((beams + 63) // 64)),
                                         "valu_instr_per_task": scan.get("SQ_INSTS_VALU", 0.0) / (agents * ((beams + 63) // 64)),
                                         "TCP_hit_frac": 1.0 - scan.get("TCP_TCC_READ_REQ_sum", 0.0) / max(scan.get("TCP_TOTAL_CACHE_ACCESSES_sum", 1.0), 1.0),
                                         "TCC_hit_frac": scan.get("TCC_HIT_sum", 0.0) / max(scan.get("TCC_REQ_sum", 1.0), 1.0)}
        json.dump(fl, open(fl_path, "w"), indent=1, sort_keys=True)
    # HBM traffic of the scan kernel for the bench's other legs (tools/gpu_r3.sh pmc: traffic_<tag>_{FETCH,WRITE}_SIZE.json)
    rec_path = os.path.join(DST, "pmc_scan.json")
    rec = json.load(open(rec_path)) if os.path.isfile(rec_path) else {}
    for leg, key in (("4096", "agents=4096,beams=1080,layout=3"), ("cfg5", "agents=65536,beams=4096,layout=3,tiles=2")):
        vals, meta = {}, {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            q = os.path.join(SRC, "traffic_%s_%s.json" % (leg, c))
            if os.path.isfile(q):
                for kern, r in json.load(open(q)).items():
                    if kern.startswith(("k_scan_rays", "k_scan_dirs")) and c in r["mean_per_dispatch"]:
                        vals[c] = r["mean_per_dispatch"][c]; meta = r
        if len(vals) == 2:
            rec[key] = {"round": tag, "csrc": meta.get("csrc"), "window": meta.get("window"), "FETCH_SIZE_KiB": vals["FETCH_SIZE"],
                        "WRITE_SIZE_KiB": vals["WRITE_SIZE"], "hbm_bytes_per_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
                        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request)"}
    json.dump(rec, open(rec_path, "w"), indent=1, sort_keys=True)
    # ... and their gather-issue floors (traffic_<tag>_VMEM.json: SQ_INSTS_VMEM_RD / _WR of the leg's scan kernel)
    fl_path = os.path.join(DST, "%s_issue_floor.json" % tag)
    if os.path.isfile(fl_path):
        fl = json.load(open(fl_path))
        for leg, key in (("4096", "agents=4096,beams=1080,layout=3"), ("cfg5", "agents=65536,beams=4096,layout=3,tiles=2")):
            q = os.path.join(SRC, "traffic_%s_VMEM.json" % leg)
            if not os.path.isfile(q):
                continue
            for kern, r in json.load(open(q)).items():
                m = r["mean_per_dispatch"]
                if kern.startswith(("k_scan_rays", "k_scan_dirs")) and "SQ_INSTS_VMEM_RD" in m and r.get("csrc") == fl.get("csrc"):
                    fl["vmem_instr_per_launch"][key] = m["SQ_INSTS_VMEM_RD"] + m.get("SQ_INSTS_VMEM_WR", 0.0)
                    fl.setdefault("pmc", {})[key] = {"kernel": kern, "waves": m.get("SQ_WAVES"), "window": r.get("window")}
        json.dump(fl, open(fl_path, "w"), indent=1, sort_keys=True)
    # other summaries of the session, as they are
    for src, dst in (("kernel_stats_4096.txt", "%s_kernel_stats_4096.txt"), ("track_scaling_65536_1080.json", "%s_track_scaling.json"),
                     ("ray_bench.txt", "%s_ray_bench.txt"), ("bench_driver_form.log", "%s_bench_driver_form.json")):
        if os.path.isfile(os.path.join(SRC, src)):
            body = open(os.path.join(SRC, src)).read()
            if csrc and '"csrc"' in body and csrc not in body:
                print("skipped %s: measured on other sources than the PMC passes (%s)" % (src, csrc))
                continue
            shutil.copyfile(os.path.join(SRC, src), os.path.join(DST, dst % tag))
    # scratch files of measurement scripts: copied only when they carry THIS session's source hash themselves (gpurun_out/ is scratch
    # that survives rounds — round 5 found round-3 files re-tagged with the new hash here)
    for src in ("latency_sizes.txt", "scan_timeline_4096.txt", "stream_scan.txt", "spec_march.txt", "finalize_wave.txt"):
        if os.path.isfile(os.path.join(SRC, src)):
            body = open(os.path.join(SRC, src)).read()
            if csrc and ("# csrc %s" % csrc) in body:
                open(os.path.join(DST, "%s_%s" % (tag, src)), "w").write(body)
    # the A/B sweeps (one JSON line per bench run) as one table
    import glob
    rows = []
    for f in sorted(glob.glob(os.path.join(SRC, "fin_*.log")) + glob.glob(os.path.join(SRC, "fin2_*.log")) + glob.glob(os.path.join(SRC, "probe_*.log")) +
                    glob.glob(os.path.join(SRC, "ray_*.log")) + glob.glob(os.path.join(SRC, "preroll_*.log"))):
        for line in open(f):
            if line.startswith("{"):
                d = json.loads(line)
                rows.append("%-28s %9.3f M agent-steps/s  %.4f ms/step  agents %6d  resets %d" % (
                    os.path.basename(f)[:-4], d["value"] / 1e6, d["ms_per_step"], d["config"]["agents_per_gpu"], d["config"]["env_resets_in_timed_region"]))
    if rows and any(not r.startswith("preroll_") for r in rows):   # (a session that only re-ran the pre-roll ramp does not replace the table)
        hdr = ("# A/B sweeps of round 3 (tools/gpu_r3.sh finalize finalize2 probes ray preroll): bench.py --only-headline --steps 300 --warmup 30,\n"
               "# experimental build, F110_EXP switches as named by the file: fin_n<agents>_f<finalize_flat>, fin_flat_l<lanes>, fin2_n<agents>_l<lanes>,\n"
               "# fin2_<workload>_pa<pair_always>, probe_n<agents>_{base,occ4,cnt} (task_order off; scan at 4 waves/SIMD; + per-env completion counter),\n"
               "# ray_n<agents>_r<ray_pass>, ray_thr<N>, ray_w<waves>, ray_tt<task_thr>, preroll_<P> (20 timed steps after P un-timed ones)\n# csrc %s\n" % csrc)
        open(os.path.join(DST, "%s_ab_sweeps.txt" % tag), "w").write(hdr + "\n".join(rows) + "\n")
    print("profiles/%s_* written" % tag)


if __name__ == "__main__":
    main()
