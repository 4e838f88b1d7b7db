"""ORACLE -- TEST INFRASTRUCTURE.  ctypes loader for oracle/libpygsd_oracle.so (pygsd_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libpygsd_oracle.so")
_lib = None


def build():
    src = os.path.join(HERE, "pygsd_oracle.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "libpygsd_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(t))


def propagate(x, edge_index, w, n_out, flow="source_to_target", mean=False):
    """numpy in / numpy out.  x float32 [N, F]; edge_index int64 [2, E]; w float32 [E] or None."""
    g, s = (0, 1) if flow == "source_to_target" else (1, 0)
    x = np.ascontiguousarray(x, dtype=np.float32)
    src = np.ascontiguousarray(edge_index[g], dtype=np.int64)
    dst = np.ascontiguousarray(edge_index[s], dtype=np.int64)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
    out = np.empty((n_out, x.shape[1]), dtype=np.float32)
    rc = lib().oracle_propagate_f32(_p(src, ctypes.c_int64), _p(dst, ctypes.c_int64), _p(w, ctypes.c_float),
                                    ctypes.c_int64(src.size), _p(x, ctypes.c_float),
                                    ctypes.c_int64(x.shape[1]), _p(out, ctypes.c_float),
                                    ctypes.c_int64(n_out), ctypes.c_int(1 if mean else 0))
    assert rc == 0
    return out


def magnetic_laplacian(edge_index, w, n, q, normalization="sym", signed=False, absolute_degree=True):
    row = np.ascontiguousarray(edge_index[0], dtype=np.int64)
    col = np.ascontiguousarray(edge_index[1], dtype=np.int64)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
    cap = 2 * row.size + n + 1
    o_row, o_col = np.empty(cap, np.int64), np.empty(cap, np.int64)
    o_re, o_im = np.empty(cap, np.float32), np.empty(cap, np.float32)
    nnz = ctypes.c_int64(0)
    rc = lib().oracle_magnetic_laplacian(_p(row, ctypes.c_int64), _p(col, ctypes.c_int64), _p(w, ctypes.c_float),
                                         ctypes.c_int64(row.size), ctypes.c_int64(n), ctypes.c_double(q),
                                         ctypes.c_int(normalization == "sym"), ctypes.c_int(bool(signed)),
                                         ctypes.c_int(bool(absolute_degree)), _p(o_row, ctypes.c_int64),
                                         _p(o_col, ctypes.c_int64), _p(o_re, ctypes.c_float),
                                         _p(o_im, ctypes.c_float), ctypes.byref(nnz))
    assert rc == 0
    k = nnz.value
    return np.stack([o_row[:k], o_col[:k]]), o_re[:k].copy(), o_im[:k].copy()


def complex_relu(re, im):
    re = np.ascontiguousarray(re, dtype=np.float32)
    im = np.ascontiguousarray(im, dtype=np.float32)
    o_r, o_i = np.empty_like(re), np.empty_like(im)
    lib().oracle_complex_relu_f32(_p(re, ctypes.c_float), _p(im, ctypes.c_float), ctypes.c_int64(re.size),
                                  _p(o_r, ctypes.c_float), _p(o_i, ctypes.c_float))
    return o_r, o_i
