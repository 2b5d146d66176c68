import json, os, subprocess, sys
ROOT = os.getcwd()
for a in (65536, 4096):
    row = []
    for blk in (64, 128, 192, 256):
        for t in ((0, 2, 4) if a == 65536 else (0,)):
            out = subprocess.run([sys.executable, "bench.py", "--only-headline", "--agents", str(a), "--scan-block", str(blk), "--scan-tasks", str(t), "--steps", "300", "--warmup", "20"], capture_output=True, text=True).stdout
            d = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
            row.append("blk %d tpw %d: %.2f M" % (blk, t, d[0]["value"] / 1e6) if d else "blk %d tpw %d: failed" % (blk, t))
    print("agents %d  " % a + "   ".join(row), flush=True)
