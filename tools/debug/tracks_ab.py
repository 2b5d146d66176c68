import glob, os, shutil, subprocess, sys, json
ROOT = os.getcwd(); pk = os.path.join(ROOT, "f1tenth_gym_amd")
shutil.copy(os.path.join(pk, "libf110_hip.so"), os.path.join(pk, "probe_zz_tree.so"))
try:
    for lib in sorted(glob.glob(os.path.join(pk, "probe_*.so"))):
        shutil.copy(lib, os.path.join(pk, "libf110_hip.so"))
        out = subprocess.run([sys.executable, "tools/debug/track_scaling.py", "65536", "1080"], capture_output=True, text=True).stdout
        rows = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
        print(os.path.basename(lib), "  ".join("%s/%s %.4f" % (r["tracks"], r["assignment"][:5], r["ms_per_step"]) for r in rows[:6]), flush=True)
finally:
    shutil.copy(os.path.join(pk, "probe_zz_tree.so"), os.path.join(pk, "libf110_hip.so")); os.remove(os.path.join(pk, "probe_zz_tree.so"))
