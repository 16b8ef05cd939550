"""Build librdgan_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m pr_disagg_radar_gan_amd.build

The .so is git-ignored but travels with the tree to the GPU box.  No torch extension
machinery: the library has a plain C ABI (include/rdgan.h) and is loaded with ctypes.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librdgan_hip.so")
SOURCES = ["rdgan_api.hip"]


def headers():
    """Every header the translation unit can include: a glob, so a new header can never be missing from the staleness check
    (round 3: four "fixes" were tested on a stale binary because a new header was not listed here)."""
    import glob
    hs = sorted(glob.glob(os.path.join(CSRC, "*.h")))
    return hs + [os.path.normpath(os.path.join(HERE, "..", "include", "rdgan.h"))]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build librdgan_hip.so)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in [os.path.join(CSRC, x) for x in SOURCES] + headers())


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-result"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
