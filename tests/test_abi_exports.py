"""CPU test: libmfgm.so loads and exports every symbol declared in include/mfgm.h (no compute calls without a GPU)."""
import ctypes
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "vi-diffusion-processes_amd", "csrc", "libmfgm.so")
HEADER = os.path.join(ROOT, "include", "mfgm.h")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="libmfgm.so not built (run __graft_entry__.build())")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mfgm_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(LIB)
    names = declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/mfgm.h but not exported: {missing}"


def test_python_binding_covers_the_header():
    sys.path.insert(0, ROOT)
    import vidp_amd
    lib = vidp_amd._lib.load()
    assert lib.mfgm_version().startswith(b"mfgm")
    unbound = set(declared_symbols()) - set(vidp_amd._lib.EXPORTS)
    assert not unbound, f"declared in include/mfgm.h but not bound in _lib.EXPORTS: {sorted(unbound)}"


def test_plan_creation_and_argument_checks_need_no_gpu():
    sys.path.insert(0, ROOT)
    import vidp_amd
    lib = vidp_amd._lib.load()
    h = ctypes.c_void_p()
    assert lib.mfgm_plan_create(64, 100000, 6, 0, 0, ctypes.byref(h)) == 0
    desc = (ctypes.c_int * 6)()
    assert lib.mfgm_plan_describe(h, desc) == 0
    assert desc[4] == 64 and desc[5] == 100000 and desc[2] * desc[1] >= 100000
    assert lib.mfgm_plan_workspace_bytes(h) > 0
    assert lib.mfgm_packed_doubles(h, 2) == desc[1] * 21 * desc[3]
    lib.mfgm_plan_destroy(h)
    # unsupported block size and bad arguments are reported, not crashed on
    assert lib.mfgm_plan_create(1, 10, 33, 0, 0, ctypes.byref(h)) == 1
    # 8 < d <= 32 takes the wide path on natural-layout arrays
    assert lib.mfgm_plan_create(2, 100, 16, 0, 0, ctypes.byref(h)) == 0
    assert lib.mfgm_packed_doubles(h, 2) == 2 * 100 * 256 and lib.mfgm_packed_doubles(h, 0) == 2 * 100 * 16
    lib.mfgm_plan_destroy(h)
    assert lib.mfgm_plan_create(0, 10, 3, 0, 0, ctypes.byref(h)) == 1
