"""CPU side of the INTEGRATION.md section B check: the documented ctypes structures (parsed from the document as it stands,
no library call) have the field order and offsets of the binding the package itself uses, and the sizes the document
quotes.  The GPU test (tests/test_gpu_integration_stub.py) executes the whole block."""
import ctypes as C
import re

import numpy as np  # noqa: F401

from test_gpu_integration_stub import _stub_source
from sgmcmc_ssm_amd import _capi


def test_documented_struct_layouts_without_a_gpu():
    src = _stub_source()
    # the two class definitions only (everything before the first library call)
    head = src[src.index("dp = C.POINTER"):src.index("assert lib.pfg_struct_size")]
    ns = {"C": C, "np": np}
    exec(head, ns)
    for mine, theirs in ((ns["pfg_problem"], _capi.Problem), (ns["pfg_result"], _capi.Result)):
        assert C.sizeof(mine) == C.sizeof(theirs)
        assert [n for n, _ in mine._fields_] == [n for n, _ in theirs._fields_]
        for n, _ in mine._fields_:
            assert getattr(mine, n).offset == getattr(theirs, n).offset, n
    sizes = [int(x) for x in re.findall(r"\((\d+) bytes\)|# (\d+) bytes", src) for x in x if x]
    assert C.sizeof(_capi.Problem) in sizes and C.sizeof(_capi.Result) in sizes, sizes
