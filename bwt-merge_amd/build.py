"""Builds libbwtm.so (the C-ABI library with the gfx950 kernels) in-tree with hipcc."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbwtm.so")
EXPERIMENTAL_LIB = os.path.join(HERE, "libbwtm_experimental.so")   # the same sources with -DBWTM_EXPERIMENTAL (include/bwtm_experimental.h)
SOURCES = ["bwtm_api.hip"]
import glob
DEPS = (["bwtm_api.hip", "bwtm_kernels.hip.h", "bwtm_device.h", "bwtm_bitmerge.h", "bwtm_view.h", os.path.join("..", "..", "include", "bwtm.h"),
         os.path.join("..", "..", "include", "bwtm_experimental.h")]
        + [os.path.relpath(f, CSRC) for f in sorted(glob.glob(os.path.join(CSRC, "kernels", "*.hip.h")) + glob.glob(os.path.join(CSRC, "api", "*.hip.h")))])


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False, diagnostics=False, experimental=False, out=None):
    """Compile for gfx950 only (no other targets, no fallbacks).  diagnostics=True adds the timing-only
    kernel variants and their bwtm_tune keys (-DBWTM_DIAGNOSTICS; never the product build); experimental=True builds
    libbwtm_experimental.so: the product sources plus the entry points of include/bwtm_experimental.h."""
    out = out or (EXPERIMENTAL_LIB if experimental else LIB)
    if not force and out in (LIB, EXPERIMENTAL_LIB) and not stale(out):
        return out
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread"] + \
          (["-DBWTM_DIAGNOSTICS"] if diagnostics else []) + (["-DBWTM_EXPERIMENTAL"] if experimental else []) + \
          ["-o", out] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return out


if __name__ == "__main__":
    build(force=True, verbose=True)
    build(force=True, verbose=True, experimental=True)
