import ctypes as C
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _lib(path):
    from csc_amd.capi import CscLib
    return CscLib(path)


@pytest.fixture(scope="session")
def orc():
    """the plain-C restatement (the checker)"""
    lib = _lib(os.path.join(ROOT, "oracle", "liborc.so"))
    lib.lib.orc_zero_alloc.restype = C.c_void_p
    lib.lib.orc_aa_alloc.restype = C.c_void_p
    return lib


@pytest.fixture(scope="session")
def zalloc(orc):
    return orc.lib.orc_zero_alloc()


@pytest.fixture(scope="session")
def ref():
    """the reference itself, built by oracle/Makefile into oracle/_ref (absent -> skip)"""
    path = os.path.join(ROOT, "oracle", "_ref", "libcsc_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libcsc_ref.so not built (needs /root/reference)")
    return _lib(path)


@pytest.fixture(scope="session")
def prod():
    """the product: HIP kernels behind the C ABI"""
    import csc_amd
    return csc_amd.load()
