"""Builds libbwtm.so (the C-ABI library with the gfx950 kernels) in-tree with hipcc."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbwtm.so")
SOURCES = ["bwtm_api.hip"]
import glob
DEPS = (["bwtm_api.hip", "bwtm_kernels.hip.h", "bwtm_device.h", os.path.join("..", "..", "include", "bwtm.h")]
        + [os.path.relpath(f, CSRC) for f in sorted(glob.glob(os.path.join(CSRC, "kernels", "*.hip.h")) + glob.glob(os.path.join(CSRC, "api", "*.hip.h")))])


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False, diagnostics=False, out=None):
    """Compile for gfx950 only (no other targets, no fallbacks).  diagnostics=True adds the timing-only
    kernel variants and their bwtm_tune keys (-DBWTM_DIAGNOSTICS; never the product build)."""
    out = out or LIB
    if not force and out == LIB and not stale():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread"] + \
          (["-DBWTM_DIAGNOSTICS"] if diagnostics else []) + ["-o", out] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
