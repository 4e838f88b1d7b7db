"""Shared launch/timing helpers of the SpMM probes.  Original note: does column-blocking a wide SpMM (F=128 -> 2 x 64 columns, F=64 -> 2 x 32) pay once the
gathered set exceeds the 256 MB Infinity Cache?  Launches the C-ABI directly with strided views."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from pytorch_geometric_signed_directed_amd import _cabi

dev = torch.device("cuda:0")
lib, P = _cabi.lib(), _cabi.ptr


def run(csr, va, vb, xa, xb, ya, yb, f0, fw, ld, hint=None):
    o = f0 * 4
    hint = csr.nnz if hint is None else hint
    if vb is None:
        _cabi.check(lib.pygsd_spmm_csr_f32(P(csr.rowptr), P(csr.col), P(va), xa.data_ptr() + o, ld, ya.data_ptr() + o, ld,
                                           None, 0, csr.n_rows, fw, 1.0, 0.0, 0, hint, None, _cabi.stream_ptr()), "spmm")
    else:
        _cabi.check(lib.pygsd_spmm2_csr_f32(P(csr.rowptr), P(csr.col), P(va), P(vb), xa.data_ptr() + o, xb.data_ptr() + o, ld,
                                            ya.data_ptr() + o, yb.data_ptr() + o, ld, None, None, 0, csr.n_rows, fw, 1.0, 0.0,
                                            hint, None, _cabi.stream_ptr()), "spmm2")


def timeit(fn, reps=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


