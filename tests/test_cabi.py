"""CPU: the C-ABI shared library loads without a GPU and exports every symbol include/pygsd_hip.h
declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pygsd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pygsd_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("pygsd_spmm_csr_f32", "pygsd_spmm2_csr_f32", "pygsd_sddmm_coo_f32", "pygsd_csr_from_coo",
                 "pygsd_version", "pygsd_last_error"):
        assert must in syms


def test_library_loads_and_exports_every_declared_symbol():
    from pytorch_geometric_signed_directed_amd import _cabi
    from pytorch_geometric_signed_directed_amd.build import build_library
    build_library()
    handle = ctypes.CDLL(_cabi.lib_path())
    for name in declared_symbols():
        assert hasattr(handle, name), f"libpygsd_hip.so does not export {name}"
    # the Python binding lists exactly the header's functions
    assert sorted(_cabi.PROTOTYPES) == declared_symbols()
    assert _cabi.lib().pygsd_version() == _cabi.ABI_VERSION
    assert _cabi.lib().pygsd_last_error() == b""


def test_argument_validation_needs_no_gpu():
    """Null / negative arguments are rejected on the host before any launch."""
    from pytorch_geometric_signed_directed_amd import _cabi
    lib = _cabi.lib()
    rc = lib.pygsd_spmm_csr_f32(None, None, None, None, 0, None, 0, None, 0, 5, 4, 1.0, 0.0, 0, 0, None, None)
    assert rc != 0 and b"null pointer" in lib.pygsd_last_error()
    rc = lib.pygsd_spmm_csr_f32(None, None, None, None, 0, None, 0, None, 0, -1, 4, 1.0, 0.0, 0, 0, None, None)
    assert rc != 0 and b"negative" in lib.pygsd_last_error()
    with pytest.raises(RuntimeError, match="negative"):
        _cabi.check(rc, "pygsd_spmm_csr_f32")
    n, ms = ctypes.c_int64(7), ctypes.c_double(1.0)
    assert lib.pygsd_prof_collect(99, ctypes.byref(n), ctypes.byref(ms)) != 0
