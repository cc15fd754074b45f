"""Build libpfgrad.so (HIP, gfx950) in-tree with hipcc.  No GPU is needed to compile.

The particle-filter kernels are instantiated in ten translation units (one per model x proposal
kernel x generator, csrc/pfg_inst_*.hip) plus the dispatcher / C ABI unit (csrc/pfgrad.hip); the
units are compiled to objects in parallel and linked into one shared library."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT_PKG = os.path.dirname(PKG_DIR)
CSRC = os.path.join(ROOT_PKG, "csrc")
REPO = os.path.dirname(ROOT_PKG)
INCLUDE = os.path.join(REPO, "include")
LIB_PATH = os.path.join(CSRC, "libpfgrad.so")
_UNITS = ["svm_prior", "garch_prior", "garch_optimal", "lgssm_prior", "lgssm_optimal"]
SOURCES = ["pfgrad.hip", "pfg_legacy_rng.hip"] + ["pfg_inst_{0}_{1}.hip".format(u, r) for u in _UNITS for r in ("device", "replay")]
HEADERS = [os.path.join(CSRC, h) for h in ("pfg_device.hpp", "pfg_math.hpp", "pfg_models.hpp", "pfg_reg_kernel.hpp",
                                           "pfg_mem_kernel.hpp", "pfg_big_kernel.hpp", "pfg_grid_kernel.hpp", "pfg_grid_dev_kernel.hpp", "pfg_grid_cdf.hpp", "pfg_elementwise.hpp", "pfg_host.hpp",
                                           "pfg_launch.hpp")] + \
          [os.path.join(INCLUDE, "pfgrad.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]


def _contract(src):
    """REPLAY kernels keep the reference's NumPy operation order (no fused multiply-add unless
    written as fma()); the device-generator kernels have no such parity to keep and fuse."""
    if not src.endswith("_device.hip"):
        return ["-ffp-contract=off"]
    flags = ["-ffp-contract=fast", "-DPFG_FAST_ALGEBRA=1"] + os.environ.get("PFG_EXTRA_DEVICE_FLAGS", "").split()
    if "_lgssm_" in os.path.basename(src) and not os.environ.get("PFG_EXTRA_DEVICE_FLAGS"):
        # measured per unit (A/B builds): the max-ILP scheduling strategy is worth 6.7 % on BASELINE config 1 (LGSSM,
        # one wave per window: 2.35 -> 2.20 ms per 16384 chains); SVM +1 %, GARCH +2 % slower with it, so only here
        flags += ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
    return flags


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


def _compile(args):
    hipcc, src, obj, verbose, extra = args
    cmd = [hipcc] + FLAGS + _contract(src) + extra + ["-I", INCLUDE, "-I", CSRC, "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed on {0}:\n{1}".format(os.path.basename(src), res.stdout))
    return obj


def build_library(force=False, verbose=False, jobs=None, tag=None, extra_flags=()):
    """Compile csrc/*.hip -> csrc/libpfgrad.so for gfx950.  -ffp-contract=off keeps the f64
    instantiation on the reference's NumPy operation order (see csrc/pfg_device.hpp).
    tag + extra_flags: a diagnostic build csrc/libpfgrad_<tag>.so (e.g. tag='stamps',
    extra_flags=['-DPFG_PHASE_STAMPS']), selected at run time with PFGRAD_LIB=<path>."""
    lib_path = LIB_PATH if tag is None else os.path.join(CSRC, "libpfgrad_{0}.so".format(tag))
    if tag is None and not force and not is_stale():
        return LIB_PATH
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build" if tag is None else "build_" + tag)
    os.makedirs(objdir, exist_ok=True)
    work = [(hipcc, os.path.join(CSRC, s), os.path.join(objdir, os.path.splitext(s)[0] + ".o"), verbose, list(extra_flags))
            for s in SOURCES]
    jobs = jobs or int(os.environ.get("PFGRAD_BUILD_JOBS", "0")) or min(len(work), os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(_compile, work))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib_path + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout)
    os.replace(lib_path + ".tmp", lib_path)
    return lib_path


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1:       # python -m sgmcmc_ssm_amd._build <tag> <flags...>: diagnostic build
        print(build_library(force=True, verbose=True, tag=sys.argv[1], extra_flags=sys.argv[2:]))
    else:
        print(build_library(force=True, verbose=True))
