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


def test_fp32_product_form_switches_are_host_state():
    """pygsd_tall_f32_form / pygsd_dense_f32_form (include/pygsd_hip.h): 0 = split form where a shape has one (default), 1 = exact
    fp32 MFMA everywhere; each call returns the form in force before it, any other argument only queries -- no GPU involved."""
    import subprocess
    import sys
    from pytorch_geometric_signed_directed_amd import _cabi
    from pytorch_geometric_signed_directed_amd.dense import set_dense_f32_exact, set_tall_f32_exact
    lib = _cabi.lib()
    for fn, setter in ((lib.pygsd_tall_f32_form, set_tall_f32_exact), (lib.pygsd_dense_f32_form, set_dense_f32_exact)):
        start = fn(-1)
        assert start in (0, 1) and fn(-1) == start          # a query changes nothing
        assert fn(1) == start and fn(-1) == 1
        assert fn(0) == 1 and fn(-1) == 0
        assert fn(7) == 0 and fn(-1) == 0                    # out of range: query only
        assert setter(True) is False and fn(-1) == 1        # the Python wrappers: previous setting as a bool
        assert setter(False) is True and fn(-1) == 0
        fn(start)
    # the environment selects the exact forms at load
    code = ("from pytorch_geometric_signed_directed_amd import _cabi; l = _cabi.lib(); "
            "print(l.pygsd_tall_f32_form(-1), l.pygsd_dense_f32_form(-1))")
    env = dict(os.environ, PYGSD_TALL_F32="exact", PYGSD_DENSE_F32="exact")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, check=True)
    assert out.stdout.split()[-2:] == ["1", "1"]
    env = {k: v for k, v in os.environ.items() if k not in ("PYGSD_TALL_F32", "PYGSD_DENSE_F32")}
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, check=True)
    assert out.stdout.split()[-2:] == ["0", "0"]
