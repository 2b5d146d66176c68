"""Compile the gfx950 extension in-tree: f1tenth_gym_amd/csrc/f110_hip.hip -> libf110_hip.so.

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the parity contract
(DESIGN.md): float64 in the reference's operation order, never contracted into FMAs.
"""
import hashlib
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG_DIR, "csrc", "f110_hip.hip")
DEPS = [SRC, os.path.join(PKG_DIR, "csrc", "f110_math.hpp"), os.path.join(PKG_DIR, "csrc", "f110_kernels.hpp"),
        os.path.join(PKG_DIR, "csrc", "f110_rng.hpp"), os.path.join(PKG_DIR, "csrc", "f110_ziggurat_tables.hpp"),
        os.path.join(os.path.dirname(PKG_DIR), "include", "f110.h")]
LIB = os.path.join(PKG_DIR, "libf110_hip.so")
LIB_EXP = os.path.join(PKG_DIR, "libf110_hip_exp.so")   # -DF110_EXPERIMENTAL: the lab (A/B variants, f110_exp_set)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def src_hash():
    """sha256 (first 16 hex digits) over the kernel sources and the ABI header, in a fixed order: the
    identity of the code a library / a profile entry / a bench line belongs to.  (A git tree hash would
    do the same, but the GPU box receives a snapshot without .git.)"""
    hsh = hashlib.sha256()
    for d in sorted(DEPS, key=os.path.basename):
        hsh.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            hsh.update(f.read())
    return hsh.hexdigest()[:16]


def find_hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.isfile(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X extension cannot be built")


def is_stale(lib=LIB):
    if not os.path.isfile(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, variant="product"):
    """Build libf110_hip.so (variant="product") or libf110_hip_exp.so ("experimental") if missing or
    older than its sources.  Returns the path."""
    lib = LIB_EXP if variant == "experimental" else LIB
    if not force and not is_stale(lib):
        return lib
    extra = os.environ.get("F110_EXTRA_HIPCC_FLAGS", "").split()   # experiments only
    if variant == "experimental":
        extra = ["-DF110_EXPERIMENTAL"] + extra
    cmd = [find_hipcc()] + HIPCC_FLAGS + ['-DF110_SRC_HASH="%s"' % src_hash()] + extra + [SRC, "-o", lib + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout)
    os.replace(lib + ".tmp", lib)
    return lib


def build_all(force=False, verbose=False):
    return [build(force, verbose, "product"), build(force, verbose, "experimental")]


if __name__ == "__main__":
    print(build_all(force=True, verbose=True))
