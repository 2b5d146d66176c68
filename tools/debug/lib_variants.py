"""A/B of product-library builds that differ in compile-time switches: each f1tenth_gym_amd/probe_*.so is copied over
libf110_hip.so in turn (the GPU box's copy of the tree is scratch) and bench.py --only-headline is run at the given sizes.
    python tools/debug/lib_variants.py 4096,65536 [bench flags...]"""
import glob, json, os, shutil, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
pk = os.path.join(ROOT, "f1tenth_gym_amd")
agents = [int(x) for x in sys.argv[1].split(",")]
extra = sys.argv[2:]
shutil.copy(os.path.join(pk, "libf110_hip.so"), os.path.join(pk, "probe_zz_tree.so"))
try:
    for lib in sorted(glob.glob(os.path.join(pk, "probe_*.so"))):
        shutil.copy(lib, os.path.join(pk, "libf110_hip.so"))
        row = []
        for a in agents:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--only-headline", "--agents", str(a), "--steps", "400", "--warmup", "20"] + extra,
                                 capture_output=True, text=True, env=dict(os.environ, F110_NO_BUILD="1"))
            ls = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
            row.append("%d: %.2f M (%.1f us)" % (a, ls[0]["value"] / 1e6, ls[0]["ms_per_step"] * 1e3) if ls else "%d: failed %s" % (a, out.stderr[-200:]))
        print("%-22s " % os.path.basename(lib) + "   ".join(row), flush=True)
finally:
    shutil.copy(os.path.join(pk, "probe_zz_tree.so"), os.path.join(pk, "libf110_hip.so"))
    os.remove(os.path.join(pk, "probe_zz_tree.so"))
