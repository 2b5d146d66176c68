"""consecutive 64-ray tasks per scan wave (f110_config.scan_tasks_per_wave) against batch size, bench workload:
    python tools/debug/tpw_sweep.py [agents,agents,...] [tpw,tpw,...]"""
import json, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
agents = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1024,2048,4096,8192,16384").split(",")]
tpws = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3,4").split(",")]
extra = sys.argv[3:]
for a in agents:
    row = []
    for t in tpws:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--only-headline", "--agents", str(a), "--scan-tasks", str(t),
                              "--steps", "400", "--warmup", "20"] + extra, capture_output=True, text=True).stdout
        d = [json.loads(l) for l in out.splitlines() if l.startswith("{")][0]
        row.append("tpw %d: %.2f M (%.1f us)" % (t, d["value"] / 1e6, d["ms_per_step"] * 1e3))
    print("agents %6d  " % a + "   ".join(row), flush=True)
