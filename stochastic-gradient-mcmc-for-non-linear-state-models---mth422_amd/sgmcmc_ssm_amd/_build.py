"""Build libpfgrad.so (HIP, gfx950) in-tree with hipcc.  No GPU is needed to compile."""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT_PKG = os.path.dirname(PKG_DIR)
CSRC = os.path.join(ROOT_PKG, "csrc")
REPO = os.path.dirname(ROOT_PKG)
INCLUDE = os.path.join(REPO, "include")
LIB_PATH = os.path.join(CSRC, "libpfgrad.so")
SOURCES = ["pfgrad.hip"]
HEADERS = [os.path.join(CSRC, "pfg_device.hpp"), os.path.join(INCLUDE, "pfgrad.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile csrc/*.hip -> csrc/libpfgrad.so for gfx950.  -ffp-contract=off keeps the f64
    instantiation on the reference's NumPy operation order (see csrc/pfg_device.hpp)."""
    if not force and not is_stale():
        return LIB_PATH
    cmd = [_hipcc(), "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17",
           "-fPIC", "-shared", "-I", INCLUDE, "-I", CSRC]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
