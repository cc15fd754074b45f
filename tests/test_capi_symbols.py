"""The C-ABI library builds, loads and exports every function include/pfgrad.h declares.
No compute calls here (no GPU in the CPU suite)."""
import ctypes
import os
import re

import pytest

from sgmcmc_ssm_amd import _capi, _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "pfgrad.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|void|int64_t|char)\s*\*?\s*(pfg_\w+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_declares_expected_entry_points():
    names = declared_functions()
    assert set(names) == set(_capi.EXPORTS), (names, _capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    if _build.is_stale():
        _build.build_library()
    lib = ctypes.CDLL(_build.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_binding_struct_sizes_and_version():
    lib = _capi.load_library()          # asserts the struct sizes against pfg_struct_size()
    assert lib.pfg_version() == 125
    assert lib.pfg_struct_size(2) == _capi.DEV_PROBLEM_DTYPE.itemsize == 376
    assert lib.pfg_struct_size(99) == -1
    assert lib.pfg_variant_name(0, 0, 0, 1, 1000) == b"wg256x4s"
    assert lib.pfg_variant_name(0, 0, 0, 1, 100) == b"wg64x2s"        # N <= 128, throughput: one wave per window
    assert lib.pfg_variant_name(0, 0, 0, 1, 200) == b"wg64x4s"         # 128 < N <= 256, device generator, many windows
    assert lib.pfg_variant_name(0, 0, 0, 0, 200) == b"wg256x1"         # ... the REPLAY units
    assert lib.pfg_variant_name(0, 0, 0, 1, 1024) == b"wg256x4s"
    assert lib.pfg_variant_name(0, 0, 0, 0, 1025) == b"mem1024"       # N > 1024: state in HBM scratch
    assert lib.pfg_variant_name(0, 0, 0, 1, 1025) == b"wg1024x4s"     # ... device generator, SVM fp64: still fits LDS (32-bit CDF)
    assert lib.pfg_variant_name(2, 1, 0, 1, 4096) == b"big4096"       # LGSSM fp64 (5 arrays) does not: fast large-N kernel
    assert lib.pfg_variant_name(2, 1, 1, 1, 4096) == b"wg1024x4s"     # ... in f32 it does
    assert lib.pfg_variant_name(0, 0, 0, 1, 10000) == b"big16384"
    assert lib.pfg_variant_name(1, 1, 0, 1, 1000) == b"wg512x2s"      # GARCH fp64 (n=2, h=4) still LDS-resident: 8 waves
    assert lib.pfg_variant_name(1, 1, 0, 0, 1000) == b"wg256x4s"      # ... the REPLAY units keep 256 x 4
    assert lib.pfg_variant_name(1, 1, 0, 0, 4000) == b"mem1024" and lib.pfg_variant_name(1, 1, 0, 1, 4000) == b"big4096"
    assert lib.pfg_variant_name(0, 0, 0, 0, 10000) == b"mem1024"
    # above the one-workgroup kernels' 16384: the whole-GPU window (tiles of 1024 particles up to 2^19, of 2048 above)
    assert lib.pfg_variant_name(0, 0, 0, 1, 20000) == b"grid1024" and lib.pfg_variant_name(0, 0, 0, 0, 500000) == b"grid1024" and lib.pfg_variant_name(0, 0, 0, 0, 1000000) == b"grid2048"
    assert lib.pfg_variant_name(0, 0, 0, 1, (1 << 22) + 1) == b"none"
    assert lib.pfg_scratch_bytes(0, 0, 1, 1000) == 0 and lib.pfg_scratch_bytes(0, 0, 1, (1 << 22) + 1) == -1
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "helpers"))
    from grid_layout import grid_layout
    for model, mname in enumerate(("svm", "garch", "lgssm")):
        for dtype, dname in enumerate(("f64", "f32")):
            for rng in (0, 1):
                for N in (16385, 100000, 1 << 19, (1 << 19) + 1, 1000000, (1 << 20) + 1, 1 << 22):
                    assert lib.pfg_scratch_bytes(model, dtype, rng, N) == grid_layout(mname, dname, N, rng == 0)["bytes"]
    assert lib.pfg_scratch_bytes(0, 0, 1, 10000) == (10000 * 9 * 8 + 16 + 255) // 256 * 256


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the product path must fail loudly, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(_capi.PfgError, match="no HIP device|no CPU fallback"):
        _capi.Context(0)
    assert "oracle" not in " ".join(open(os.path.join(os.path.dirname(_capi.__file__), f)).read()
                                    for f in os.listdir(os.path.dirname(_capi.__file__)) if f.endswith(".py")
                                    ).replace("oracle fixtures", "")


@pytest.mark.gpu
def test_plain_c_consumer_gpu(tmp_path):
    """The same C program on an MI355X: pfg_create + pfg_run from plain C reproduce the REFERENCE's numbers
    for the traced case pf_trace.npz:c0 (tests/c_abi/known_answer.h) at rtol 1e-9; the program returns non-zero
    and this test fails if log-likelihood or any gradient component differs."""
    out = _build_and_run_c_consumer(tmp_path)
    assert "run rc=0" in out and "known answer ok" in out and "differs" not in out, out


def test_plain_c_consumer(tmp_path):
    """include/pfgrad.h compiles as C99 with gcc and a C program links against libpfgrad.so
    (without a GPU it must report the missing device loudly)."""
    out = _build_and_run_c_consumer(tmp_path)
    assert "run rc=0" in out or "create failed as expected" in out, out


def _build_and_run_c_consumer(tmp_path):
    import shutil
    import subprocess
    if _build.is_stale():
        _build.build_library()
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    exe = str(tmp_path / "abi_check")
    libdir = os.path.dirname(_build.LIB_PATH)
    cmd = [gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "tests", "c_abi"),
           os.path.join(ROOT, "tests", "c_abi", "abi_check.c"), "-o", exe,
           "-L", libdir, "-lpfgrad", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    return run.stdout
