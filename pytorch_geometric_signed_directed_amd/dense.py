"""Host wrappers of the fused MFMA dense stage (csrc/dense.hip; include/pygsd_hip.h
pygsd_magnetic_dense_*) and the single autograd node of a whole MagNetConv / MSConv layer."""
import ctypes
import os
from ctypes import c_void_p
from typing import List, Optional, Sequence, Tuple

import torch

from . import _cabi
from ._cabi import check, ptr, stream_ptr
from .sparse import _spmm2_raw

Tensor = torch.Tensor


def dense_supported(f_in: int, f_out: int, k1: int) -> bool:
    return bool(_cabi.lib().pygsd_magnetic_dense_supported(int(f_in), int(f_out), int(k1)))


def _ptr_array(ts: Sequence[Tensor]):
    return (c_void_p * len(ts))(*[t.data_ptr() for t in ts])


class PieceOperand:
    """The last term's operand pair of the dense stage living in exchange buffers: `buffer` (fp32, kept alive by whoever holds
    this), the two groups' first elements `off_a` / `off_b` (floats into it) and the `_cabi.PieceLayout` that places the rows."""

    def __init__(self, buffer: Tensor, off_a: int, off_b: int, layout):
        self.buffer, self.off_a, self.off_b, self.layout = buffer, int(off_a), int(off_b), layout

    def ptrs(self):
        return self.buffer.data_ptr() + 4 * self.off_a, self.buffer.data_ptr() + 4 * self.off_b


def _ptr_array_with(ts: Sequence[Tensor], last: Optional[int]):
    ptrs = [t.data_ptr() for t in ts] + ([] if last is None else [last])
    return (c_void_p * len(ptrs))(*ptrs)


def dense_fwd_raw(a: List[Tensor], b: List[Tensor], weight: Tensor, bias: Optional[Tensor],
                  last_in: Optional[PieceOperand] = None) -> Tuple[Tensor, Tensor]:
    """out_real = sum_k (a_k - b_k) W_k + bias ; out_imag = sum_k (a_k + b_k) W_k + bias.
    last_in (sharded layers): the LAST term's operands are read through a piece layout (pygsd_magnetic_dense_fwd_pieces_f32) --
    `a` / `b` then hold the k1 - 1 terms before it."""
    k1, f_in, f_out = weight.shape
    for t in list(a) + list(b) + [weight]:
        if t.dtype != torch.float32:
            raise TypeError(f"the fused dense stage computes in float32; got {t.dtype}")
    a = [_cabi.c16(t) for t in a]
    b = [_cabi.c16(t) for t in b]
    if len(a) != (k1 if last_in is None else k1 - 1) or len(b) != len(a):
        raise ValueError(f"{k1} Chebyshev terms expected, got {len(a)} (+ a piece operand: {last_in is not None})")
    n = a[0].size(0)
    w = weight.detach().contiguous()
    out_r = torch.empty((n, f_out), dtype=torch.float32, device=w.device)
    out_i = torch.empty_like(out_r)
    bias_c = None if bias is None else bias.detach().contiguous()
    with _cabi.on_device(w.device):
        if last_in is None:
            check(_cabi.lib().pygsd_magnetic_dense_fwd_f32(_ptr_array(a), _ptr_array(b), k1, ptr(w), ptr(bias_c),
                                                           ptr(out_r), ptr(out_i), n, f_in, f_out, stream_ptr()),
                  "pygsd_magnetic_dense_fwd_f32")
        else:
            pa, pb = last_in.ptrs()
            lay = last_in.layout.struct()
            check(_cabi.lib().pygsd_magnetic_dense_fwd_pieces_f32(_ptr_array_with(a, pa), _ptr_array_with(b, pb), k1, ptr(w),
                                                                  ptr(bias_c), ptr(out_r), ptr(out_i), n, f_in, f_out,
                                                                  ctypes.byref(lay), stream_ptr()),
                  "pygsd_magnetic_dense_fwd_pieces_f32")
    return out_r, out_i


def gather_pieces(src: PieceOperand, n_rows: int, width: int, z: Optional[Sequence[Tensor]] = None, groups: int = 2):
    """[(z_g +) group g for g < groups] of a product that lives in a return exchange's receive buffer, as [n_rows, width] rows in
    local order (pygsd_gather_pieces_f32): the merge, with the addend of the adjoint's steps (dT_{k-1} + S^T dT_k) folded into the
    same pass.  Group g's rows start `g * (off_b - off_a)` floats behind group 0's."""
    dev = src.buffer.device
    outs = [torch.empty((n_rows, width), dtype=torch.float32, device=dev) for _ in range(groups)]
    zs = None
    if z is not None:
        zc = [t.contiguous() for t in z]
        zs = _ptr_array(zc)
    step = 4 * (src.off_b - src.off_a)
    first = src.buffer.data_ptr() + 4 * src.off_a
    lay = src.layout.struct()
    with _cabi.on_device(dev):
        check(_cabi.lib().pygsd_gather_pieces_f32((c_void_p * groups)(*[first + g * step for g in range(groups)]), ctypes.byref(lay),
                                                  zs, width, _ptr_array(outs), width, groups, n_rows, width, stream_ptr()),
              "pygsd_gather_pieces_f32")
    return outs


def dense_bwd_raw(a: List[Tensor], b: List[Tensor], weight: Tensor, g_r: Tensor, g_i: Tensor,
                  rows: Optional[int] = None, last_in: Optional[PieceOperand] = None, last_out: Optional[PieceOperand] = None):
    """-> (da list, db list, dW [k1, f_in, f_out], dbias [f_out]).
    rows: only the first `rows` rows enter (the sharded layer's real rows; the pad rows behind them are not part of
    the graph): dW / dbias sum over those rows, da / db keep the full height with zero rows behind.
    last_in / last_out (sharded layers, pygsd_magnetic_dense_bwd_pieces_f32): the last term's operands are read through a piece
    layout (`a` / `b` hold the k1 - 1 terms before it) / its gradients are stored through one -- straight into the send buffers
    of the propagate that takes them; da[k1 - 1] / db[k1 - 1] are None then."""
    k1, f_in, f_out = weight.shape
    a = [_cabi.c16(t) for t in a]
    b = [_cabi.c16(t) for t in b]
    if len(a) != (k1 if last_in is None else k1 - 1) or len(b) != len(a):
        raise ValueError(f"{k1} Chebyshev terms expected, got {len(a)} (+ a piece operand: {last_in is not None})")
    if g_r.size(0) > 1 and g_r.stride(0) == 0 and g_i.stride(0) == 0:
        # one row broadcast to every node (the gradient of a loss that sums over the nodes arrives as an expanded
        # tensor): hand the kernel that row with a zero row stride instead of materialising two [N, F] copies
        g_r, g_i, ldg = _cabi.c16(g_r[:1]), _cabi.c16(g_i[:1]), 0
    else:
        g_r, g_i, ldg = _cabi.c16(g_r), _cabi.c16(g_i), f_out
    n_full = a[0].size(0)
    n = n_full if rows is None else int(rows)
    dev = weight.device
    w = weight.detach().contiguous()
    k_plain = k1 if last_out is None else k1 - 1
    da = [torch.empty((n_full, f_in), dtype=torch.float32, device=dev) for _ in range(k_plain)]
    db = [torch.empty((n_full, f_in), dtype=torch.float32, device=dev) for _ in range(k_plain)]
    if n < n_full:
        for t in da + db:
            t[n:].zero_()
    if last_out is not None:
        da, db = da + [None], db + [None]          # (the send buffers' pad rows hold zeros: nothing ever writes them otherwise)
    if n == 0:                                     # a shard without real rows contributes nothing
        return da, db, torch.zeros((k1, f_in, f_out), dtype=torch.float32, device=dev), \
            torch.zeros(f_out, dtype=torch.float32, device=dev)
    dw = torch.empty((k1, f_in, f_out), dtype=torch.float32, device=dev)
    dbias = torch.empty(f_out, dtype=torch.float32, device=dev)
    lib = _cabi.lib()
    with _cabi.on_device(dev):
        need = ctypes.c_size_t(0)
        check(lib.pygsd_magnetic_dense_bwd_workspace(n, f_in, f_out, k1, ctypes.byref(need)),
              "pygsd_magnetic_dense_bwd_workspace")
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        if last_in is None and last_out is None:
            check(lib.pygsd_magnetic_dense_bwd_f32(_ptr_array(a), _ptr_array(b), k1, ptr(w), ptr(g_r), ptr(g_i), ldg,
                                                   _ptr_array(da), _ptr_array(db), ptr(dw), ptr(dbias), n, f_in,
                                                   f_out, ptr(ws), need.value, stream_ptr()),
                  "pygsd_magnetic_dense_bwd_f32")
        else:
            pin = (None, None) if last_in is None else last_in.ptrs()
            pout = (None, None) if last_out is None else last_out.ptrs()
            lin = None if last_in is None else last_in.layout.struct()
            lout = None if last_out is None else last_out.layout.struct()
            check(lib.pygsd_magnetic_dense_bwd_pieces_f32(
                _ptr_array_with(a, pin[0]), _ptr_array_with(b, pin[1]), k1, ptr(w), ptr(g_r), ptr(g_i), ldg,
                _ptr_array_with(da[:k_plain], pout[0]), _ptr_array_with(db[:k_plain], pout[1]), ptr(dw), ptr(dbias), n, f_in, f_out,
                ptr(ws), need.value, None if lin is None else ctypes.byref(lin), None if lout is None else ctypes.byref(lout),
                stream_ptr()), "pygsd_magnetic_dense_bwd_pieces_f32")
    return da, db, dw, dbias


class FixedSpmm2(torch.autograd.Function):
    """(ya, yb) = alpha * (S_r^T-chain step on xa, S_i^T-chain step on xb) + beta * (za, zb) for an operator
    whose values carry no gradient: forward on the by-target values, backward on the by-source values of
    the shared CSR."""

    @staticmethod
    def forward(ctx, xa, xb, za, zb, op, alpha: float, beta: float):
        ya, yb = _spmm2_raw(op.csr, op.values_fwd[0], op.values_fwd[1], xa, xb, za, zb, alpha, beta)
        ctx.op, ctx.alpha, ctx.beta = op, alpha, beta
        return ya, yb

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, ga, gb):
        op = ctx.op
        gxa = gxb = gza = gzb = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gxa, gxb = _spmm2_raw(op.csr, op.values_bwd[0], op.values_bwd[1], ga.contiguous(), gb.contiguous(),
                                  None, None, ctx.alpha, 0.0)
        if ctx.needs_input_grad[2]:
            gza = ga * ctx.beta
        if ctx.needs_input_grad[3]:
            gzb = gb * ctx.beta
        return gxa, gxb, gza, gzb, None, None, None


_FUSED_K1 = os.environ.get("PYGSD_FUSE_K1", "0") == "1"


def set_fused_k1(on: bool) -> bool:
    """K = 1 layers with 64 -> 64 features: run the forward dense stage in the dual SpMM's epilogue
    (pygsd_spmm2_k1_dense_f32) instead of as its own MFMA pass.  Off by default (PYGSD_FUSE_K1=1 turns it on): measured
    both ways, DESIGN.md section 5.  Returns the previous setting."""
    global _FUSED_K1
    prev, _FUSED_K1 = _FUSED_K1, bool(on)
    return prev


def spmm2_k1_dense_raw(csr, va: Tensor, vb: Tensor, x_real: Tensor, x_imag: Tensor, weight: Tensor, bias: Optional[Tensor]):
    """-> (T1_real, T1_imag, out_real, out_imag) of a K = 1, 64 -> 64 magnetic layer in ONE launch."""
    n = csr.n_rows
    xa, xb = _cabi.c16(x_real), _cabi.c16(x_imag)
    w = weight.detach().contiguous()
    bd = None if bias is None else bias.detach().contiguous()
    ta, tb = torch.empty_like(xa), torch.empty_like(xb)
    out_r, out_i = torch.empty((n, 64), dtype=torch.float32, device=xa.device), torch.empty((n, 64), dtype=torch.float32, device=xa.device)
    with _cabi.on_device(xa.device):
        check(_cabi.lib().pygsd_spmm2_k1_dense_f32(ptr(csr.rowptr), ptr(csr.col), ptr(va), ptr(vb), ptr(xa), ptr(xb), 64, ptr(ta),
                                                   ptr(tb), 64, ptr(w), ptr(bd), ptr(out_r), ptr(out_i), 64, n, csr.nnz,
                                                   stream_ptr()), "pygsd_spmm2_k1_dense_f32")
    return ta, tb, out_r, out_i


class MagneticConvFunction(torch.autograd.Function):
    """One autograd node for a whole MagNetConv / MSConv layer with fixed operator values:
    K fused dual-value SpMMs (Chebyshev recurrence in the kernel epilogue) + one fused MFMA dense
    pass forward; one fused MFMA dense pass + K SpMMs over the by-source values backward (the adjoint
    recurrence gT_{k-1} += 2 S^T gT_k, gT_{k-2} -= gT_k and the final gX = gT_0 + S^T gT_1 ride on
    the SpMM's beta*Z epilogue)."""

    @staticmethod
    def forward(ctx, x_real, x_imag, weight, bias, op):
        k1 = weight.size(0)
        csr, (vr, vi) = op.csr, op.values_fwd
        if (_FUSED_K1 and k1 == 2 and weight.size(1) == 64 and weight.size(2) == 64 and x_real.dtype == torch.float32
                and csr.n_rows == csr.n_cols == x_real.size(0) and csr.nnz > 0 and csr.hubs() is None):
            xr, xi = _cabi.c16(x_real), _cabi.c16(x_imag)
            t1r, t1i, out_r, out_i = spmm2_k1_dense_raw(csr, vr, vi, xr, xi, weight, bias)
            ctx.op, ctx.k1, ctx.has_bias = op, k1, bias is not None
            ctx.save_for_backward(weight, xr, t1r, xi, t1i)
            return out_r, out_i
        ta, tb = [_cabi.c16(x_real)], [_cabi.c16(x_imag)]
        for k in range(1, k1):
            if k == 1:
                ya, yb = _spmm2_raw(csr, vr, vi, ta[0], tb[0], None, None, 1.0, 0.0)
            else:
                ya, yb = _spmm2_raw(csr, vr, vi, ta[k - 1], tb[k - 1], ta[k - 2], tb[k - 2], 2.0, -1.0)
            ta.append(ya)
            tb.append(yb)
        out_r, out_i = dense_fwd_raw(ta, tb, weight, bias)
        ctx.op, ctx.k1, ctx.has_bias = op, k1, bias is not None
        ctx.save_for_backward(weight, *ta, *tb)
        return out_r, out_i

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_r, g_i):
        saved = ctx.saved_tensors
        weight, k1 = saved[0], ctx.k1
        ta, tb = list(saved[1:1 + k1]), list(saved[1 + k1:1 + 2 * k1])
        da, db, dw, dbias = dense_bwd_raw(ta, tb, weight, g_r, g_i)
        gx_r = gx_i = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            csr, (vr, vi) = ctx.op.csr, ctx.op.values_bwd
            for k in range(k1 - 1, 1, -1):
                da[k - 1], db[k - 1] = _spmm2_raw(csr, vr, vi, da[k], db[k], da[k - 1], db[k - 1], 2.0, 1.0)
                da[k - 2].sub_(da[k])
                db[k - 2].sub_(db[k])
            if k1 > 1:
                gx_r, gx_i = _spmm2_raw(csr, vr, vi, da[1], db[1], da[0], db[0], 1.0, 1.0)
            else:
                gx_r, gx_i = da[0], db[0]
        return gx_r, gx_i, dw, (dbias if ctx.has_bias else None), None


# ------------------------------------------------------------------------------------------------
# tall-skinny linear maps of the non-magnetic layers (csrc/tall.hip)
# ------------------------------------------------------------------------------------------------
_TALL_DTYPES = {torch.float32: 0, torch.bfloat16: 1}
_TALL_KERNELS = os.environ.get("PYGSD_LIBRARY_GEMMS", "0") != "1"


def set_tall_kernels(on: bool) -> bool:
    """Route the tall linear maps / column sums through csrc/tall.hip (default) or through library GEMMs and torch
    reductions (measurement, A/B).  Returns the previous setting."""
    global _TALL_KERNELS
    prev, _TALL_KERNELS = _TALL_KERNELS, bool(on)
    return prev


def set_dense_f32_exact(on: bool) -> bool:
    """The magnetic dense stage's products (forward and backward) as fmaf chains on the exact fp32 MFMA (True) or, at the shapes
    that have one, in the split form on the bf16 matrix pipe (False, the default; include/pygsd_hip.h, pygsd_dense_f32_form).
    `PYGSD_DENSE_F32=exact` selects the exact form at load.  Returns the previous setting; process-wide."""
    return bool(_cabi.lib().pygsd_dense_f32_form(1 if on else 0))


def set_tall_f32_exact(on: bool) -> bool:
    """fp32 tall products as an fmaf chain per output on the exact fp32 MFMA (True) or, where the shape allows, by three-way
    bf16 splitting on the bf16 matrix pipe (False, the default: faster and closer to the float64 product, but not bitwise an
    fp32 summation order -- include/pygsd_hip.h, pygsd_tall_f32_form).  `PYGSD_TALL_F32=exact` selects the exact form at load.
    Returns the previous setting; process-wide."""
    return bool(_cabi.lib().pygsd_tall_f32_form(1 if on else 0))


def _row_major16(t: Tensor) -> Tensor:
    """t as the kernels address it: unit column stride, 16-byte aligned rows (column slices of a wider matrix qualify)."""
    vec = 16 // t.element_size()
    if t.stride(1) == 1 and t.stride(0) % vec == 0 and t.stride(0) >= t.size(1) and t.data_ptr() % 16 == 0:
        return t
    return _cabi.c16(t)


def tall_product(segments: Sequence[Tensor], w: Tensor, transposed: bool = False, bias: Optional[Tensor] = None,
                 splits: Optional[Sequence[int]] = None):
    """[X_0 | X_1 | ...] @ W (+ bias) for tall segments [N, K_s] in ONE pass (pygsd_tall_linear), without concatenating.
    W: [K, F_out] with K = sum of the segment widths, or [F_out, K] with transposed=True (the input gradient
    [g | dP] W^T of a layer whose forward weight is W).  splits: column widths summing to F_out -- the result comes back
    as one CONTIGUOUS matrix per width (each consumer then gathers whole rows) instead of one [N, F_out] matrix.
    Shapes the kernel does not take run as library GEMMs."""
    x0 = segments[0]
    n, dtype = x0.size(0), x0.dtype
    k_total = sum(int(t.size(1)) for t in segments)
    f_out = int(w.size(0) if transposed else w.size(1))
    if (w.size(1) if transposed else w.size(0)) != k_total:
        raise ValueError(f"segments are {k_total} columns wide in total, W is {tuple(w.shape)} (transposed={transposed})")
    if splits is not None and sum(splits) != f_out:
        raise ValueError(f"splits {tuple(splits)} do not add up to {f_out} output columns")
    code = _TALL_DTYPES.get(dtype)
    kw = 32 if dtype == torch.bfloat16 else 16
    fused = (_TALL_KERNELS and code is not None and x0.is_cuda and len(segments) <= 4 and w.dtype == dtype
             and all(t.dim() == 2 and t.dtype == dtype and t.size(0) == n and t.size(1) % kw == 0 for t in segments)
             and (bias is None or bias.dtype == dtype)
             and (splits is None or (len(splits) <= 8 and all(c > 0 and c % kw == 0 for c in splits)))
             and bool(_cabi.lib().pygsd_tall_linear_supported(code, k_total, f_out)))
    if not fused:
        y = _gemm_fallback(segments, w, transposed, bias, k_total, f_out)
        return y if splits is None else list(y.split(list(splits), dim=1))
    segs = [_row_major16(t.detach()) for t in segments]
    wd = w.detach()
    if wd.stride(1) != 1:
        wd = wd.contiguous()
    bd = None if bias is None else bias.detach().contiguous()
    outs = [torch.empty((n, c), dtype=dtype, device=x0.device) for c in (splits if splits is not None else (f_out,))]
    k, m = len(segs), len(outs)
    xs = (c_void_p * k)(*[t.data_ptr() for t in segs])
    lds = (ctypes.c_int64 * k)(*[t.stride(0) for t in segs])
    wid = (ctypes.c_int32 * k)(*[t.size(1) for t in segs])
    ys = (c_void_p * m)(*[t.data_ptr() for t in outs])
    ldy = (ctypes.c_int64 * m)(*[t.stride(0) for t in outs])
    owid = (ctypes.c_int32 * m)(*[t.size(1) for t in outs])
    with _cabi.on_device(x0.device):
        check(_cabi.lib().pygsd_tall_linear(xs, lds, wid, k, ptr(wd), wd.stride(0), 1 if transposed else 0, ptr(bd), ys, ldy,
                                            owid, m, n, code, stream_ptr()), "pygsd_tall_linear")
    return outs[0] if splits is None else outs


def column_sums(x: Tensor) -> Tensor:
    """x.sum(0) of a tall [N, F] matrix, accumulated in fp32 (pygsd_column_sums); result in x.dtype.  A row broadcast to
    every node (stride 0: the gradient of a sum over the nodes) is N times that row."""
    code = _TALL_DTYPES.get(x.dtype)
    vec = 8 if x.dtype == torch.bfloat16 else 4
    if not (_TALL_KERNELS and code is not None and x.is_cuda and x.dim() == 2 and x.size(1) % vec == 0
            and 0 < x.size(1) <= 256 * vec // 2 and x.size(0) > 0):
        if _TALL_KERNELS and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.size(0) > 0 and x.size(1) > 0:
            # widths the 16-byte-row kernel does not take (a 10-class head): 1^T x through the generic HIP GEMM, whose
            # split reduction adds the row ranges in a fixed order (the row of ones is a stride-0 view, never materialised)
            ones = torch.ones(1, dtype=torch.float32, device=x.device).expand(1, x.size(0))
            return gemm(ones, x).view(-1)
        if _TALL_KERNELS and x.is_cuda and x.dim() == 2 and x.dtype == torch.bfloat16 and x.size(0) > 0 and x.size(1) > 0:
            ones = torch.ones(1, dtype=torch.bfloat16, device=x.device).expand(1, x.size(0))      # (the same, bf16 storage)
            return gemm_bf16(ones, x, out_dtype=torch.float32).view(-1).to(x.dtype)
        if x.is_cuda and x.dim() == 2 and x.size(0) >= 4096:          # a tall reduction outside the kernels' shapes
            _cabi.note_library_route("column_sums", f"{tuple(x.shape)} {x.dtype}")
        return x.sum(0)
    if x.size(0) > 1 and x.stride(0) == 0:
        return (x[0].float() * x.size(0)).to(x.dtype)
    xd = _row_major16(x.detach())
    n, f = xd.shape
    out = torch.empty(f, dtype=torch.float32, device=x.device)
    lib = _cabi.lib()
    with _cabi.on_device(x.device):
        need = ctypes.c_size_t(0)
        check(lib.pygsd_column_sums_workspace(n, f, code, ctypes.byref(need)), "pygsd_column_sums_workspace")
        ws = torch.empty(need.value, dtype=torch.uint8, device=x.device)
        check(lib.pygsd_column_sums(ptr(xd), xd.stride(0), n, f, code, ptr(out), ptr(ws), need.value, stream_ptr()),
              "pygsd_column_sums")
    return out if x.dtype == torch.float32 else out.to(x.dtype)


def column_sums_of(tensors: Sequence[Optional[Tensor]]) -> List[Optional[Tensor]]:
    """column_sums of each tensor, computing ONE reduction for tensors that are the same memory (the three branches of an
    inception block summed by the model receive one upstream gradient)."""
    done, out = {}, []
    for t in tensors:
        if t is None:
            out.append(None)
            continue
        key = (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype)
        if key not in done:
            done[key] = column_sums(t)
        out.append(done[key])
    return out


class _TallLinear(torch.autograd.Function):
    """y = x @ W (+ b) for a tall x [N, F_in] (N ~ 10^5..10^6, F ~ 10^1..10^2): forward and dX by tall_product
    (csrc/tall.hip), the weight gradient dW = x^T g by tall_gram (csrc/gram.hip) -- a reduction over N with only
    F_in x F_out outputs, which rocBLAS runs on a handful of CUs (measured 1.8 ms at N = 10^6, F = 64).  Shapes the MFMA
    kernels do not tile take the generic HIP GEMM (fp32) or, counted and announced, library GEMMs (batched split-K over
    4096-row slabs for the weight gradient)."""
    SLAB = 4096

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return tall_product([x], weight, False, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = tall_product([g], weight, True)
        if ctx.needs_input_grad[1]:
            gw = tall_gram(x, g)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = column_sums(g)
        return gx, gw, gb


def _gemm_fallback(segments, w, transposed, bias, k_total, f_out):
    """[X_0 | X_1 | ...] W (+ bias) for shapes pygsd_tall_linear does not tile: the generic HIP GEMM (fp32 or bf16 storage,
    any shape), else -- counted and announced (_cabi.note_library_route) -- library GEMMs."""
    x0 = segments[0]
    dt = x0.dtype
    if _TALL_KERNELS and x0.is_cuda and dt in (torch.float32, torch.bfloat16) and w.dtype == dt \
            and all(t.dim() == 2 and t.dtype == dt for t in segments) and (bias is None or bias.dtype == dt):
        y, at = None, 0
        if dt == torch.float32:
            for t in segments:
                blk = w[:, at:at + t.size(1)].t() if transposed else w[at:at + t.size(1)]
                y = gemm(t, blk, bias=bias if y is None else None, out=y, accumulate=y is not None)
                at += t.size(1)
            return y
        # bf16: the segments accumulate in an fp32 buffer; the last product adds it, the bias, and rounds to bf16 once
        for i, t in enumerate(segments):
            blk = w[:, at:at + t.size(1)].t() if transposed else w[at:at + t.size(1)]
            last = i == len(segments) - 1
            y = gemm_bf16(t, blk, bias=bias if last else None, addend=y, out_dtype=dt if last else torch.float32)
            at += t.size(1)
        return y
    if x0.is_cuda:                    # (CPU tensors only reach this through the gloo restatements of the sharded tests)
        _cabi.note_library_route("tall_product", f"{tuple(x0.shape)} {x0.dtype} x K={k_total} -> {f_out}")
    y, at = None, 0
    for t in segments:
        blk = w[:, at:at + t.size(1)].t() if transposed else w[at:at + t.size(1)]
        if y is None:
            y = torch.addmm(bias, t, blk) if bias is not None else t @ blk
        else:
            y.addmm_(t, blk)
        at += t.size(1)
    return y


def gemm_bf16(a: Tensor, b: Tensor, bias: Optional[Tensor] = None, addend: Optional[Tensor] = None,
              out_dtype: torch.dtype = torch.bfloat16) -> Tensor:
    """a [M, K] @ b [K, N] (+ bias [N]) (+ addend [M, N] fp32) for bf16 operands of ANY shapes and strides (pygsd_gemm_bf16):
    exact products, fp32 fmaf chains, the result in fp32 or rounded to bf16 once."""
    m, k = a.shape
    n = b.size(1)
    a, b = a.detach(), b.detach()
    out = torch.empty((m, n), dtype=out_dtype, device=a.device)
    if m == 0 or n == 0:
        return out
    z = None if addend is None else addend.detach()
    if z is not None and (z.dtype != torch.float32 or z.stride(1) != 1):
        z = z.float().contiguous()
    lib = _cabi.lib()
    with _cabi.on_device(a.device):
        need = ctypes.c_size_t(0)
        check(lib.pygsd_gemm_f32_workspace(m, n, k, ctypes.byref(need)), "pygsd_gemm_f32_workspace")
        ws = torch.empty(max(need.value, 16), dtype=torch.uint8, device=a.device)
        check(lib.pygsd_gemm_bf16(ptr(a), a.stride(0), a.stride(1), ptr(b), b.stride(0), b.stride(1),
                                  ptr(None if bias is None else bias.detach().contiguous()), ptr(out), out.stride(0),
                                  1 if out_dtype == torch.float32 else 0, ptr(z), 0 if z is None else z.stride(0), m, n, k,
                                  ptr(ws), need.value, stream_ptr()), "pygsd_gemm_bf16")
    return out


def gemm(a: Tensor, b: Tensor, bias: Optional[Tensor] = None, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """a [M, K] @ b [K, N] (+ bias [N]) in exact fp32 (fmaf chains) for ANY shapes and strides (views, transposes):
    pygsd_gemm_f32, the generic tiled kernel behind the shapes the MFMA kernels do not take (odd widths, K > 256, the
    2879-wide first layer of BASELINE config 1).  accumulate: add onto `out`.  A tall reduction (K >> M, N: a weight
    gradient) is split over the blocks and summed in a fixed order."""
    m, k = a.shape
    n = b.size(1)
    a, b = a.detach(), b.detach()
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    if m == 0 or n == 0:
        return out
    lib = _cabi.lib()
    with _cabi.on_device(a.device):
        need = ctypes.c_size_t(0)
        check(lib.pygsd_gemm_f32_workspace(m, n, k, ctypes.byref(need)), "pygsd_gemm_f32_workspace")
        ws = torch.empty(max(need.value, 16), dtype=torch.uint8, device=a.device)
        check(lib.pygsd_gemm_f32(ptr(a), a.stride(0), a.stride(1), ptr(b), b.stride(0), b.stride(1),
                                 ptr(None if bias is None else bias.detach().contiguous()), ptr(out), out.stride(0), m, n, k,
                                 1 if accumulate else 0, ptr(ws), need.value, stream_ptr()), "pygsd_gemm_f32")
    return out


class _Matmul(torch.autograd.Function):
    """a @ b through the generic HIP GEMM, with its two gradient products."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return gemm(a, b) if a.dtype == torch.float32 else gemm_bf16(a, b)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        mm = gemm if a.dtype == torch.float32 else gemm_bf16
        ga = mm(g, b.t()) if ctx.needs_input_grad[0] else None
        gb = mm(a.t(), g) if ctx.needs_input_grad[1] else None
        return ga, gb


def matmul(a: Tensor, b: Tensor) -> Tensor:
    """a [M, K] @ b [K, N] for fp32 device matrices of any shape / stride through pygsd_gemm_f32 (differentiable): the
    small or odd-shaped products around the path -- cluster flows P^T (A P), volumes, 1..6-column read-outs -- that the
    libraries run at ~100 GB/s (rocBLAS gemv on a 5 * 10^5 x 32 operand: 4.5 ms).  Other inputs: torch.matmul, counted."""
    if (_TALL_KERNELS and a.is_cuda and a.dim() == 2 and b.dim() == 2 and a.dtype == b.dtype
            and a.dtype in (torch.float32, torch.bfloat16)):
        return _Matmul.apply(a, b)
    if a.is_cuda and max(a.size(-2), a.size(-1)) >= 4096:
        _cabi.note_library_route("matmul", f"{tuple(a.shape)} @ {tuple(b.shape)} {a.dtype}")
    return torch.matmul(a, b)


def _gram_chunks(width: int, cap: int) -> int:
    """Column chunks pygsd_tall_gram cuts a segment into (8 / 4 / 2 / 1 tiles of 16 columns, none above `cap`)."""
    left, count = width // 16, 0
    while left > 0:
        left -= 8 if (left >= 8 and cap >= 8) else 4 if left >= 4 else 2 if left >= 2 else 1
        count += 1
    return count


def tall_gram(xs, gs) -> Tensor:
    """[X_0 | X_1 | ...]^T [G_0 | G_1 | ...] for tall segments (N ~ 10^5..10^7 rows): the weight gradients dW = x^T dY of
    the tall linear maps in ONE pass over the operands (pygsd_tall_gram: coalesced row loads, LDS transposition, MFMA,
    deterministic partial sums) -- every segment pair's block of the result side by side.  Tensors or lists of tensors."""
    xs = [xs] if isinstance(xs, Tensor) else list(xs)
    gs = [gs] if isinstance(gs, Tensor) else list(gs)
    x0 = xs[0]
    n, dtype = x0.size(0), x0.dtype
    code = _TALL_DTYPES.get(dtype)
    k_total, f_total = sum(int(t.size(1)) for t in xs), sum(int(t.size(1)) for t in gs)
    fused = (_TALL_KERNELS and code is not None and x0.is_cuda and len(xs) <= 8 and len(gs) <= 8
             and all(t.dim() == 2 and t.dtype == dtype and t.size(0) == n and t.size(1) % 16 == 0 and t.size(1) > 0
                     for t in xs + gs)
             and sum(_gram_chunks(t.size(1), 4) for t in xs) <= 16 and sum(_gram_chunks(t.size(1), 8) for t in gs) <= 16)
    if fused:
        xd, gd = [_row_major16(t.detach()) for t in xs], [_row_major16(t.detach()) for t in gs]
        out = torch.empty((k_total, f_total), dtype=torch.float32, device=x0.device)
        arr = lambda ts, fn, ct: (ct * len(ts))(*[fn(t) for t in ts])  # noqa: E731
        lib = _cabi.lib()
        with _cabi.on_device(x0.device):
            need = ctypes.c_size_t(0)
            check(lib.pygsd_tall_gram_workspace(n, k_total, f_total, code, ctypes.byref(need)), "pygsd_tall_gram_workspace")
            ws = torch.empty(max(need.value, 16), dtype=torch.uint8, device=x0.device)
            check(lib.pygsd_tall_gram(arr(xd, lambda t: t.data_ptr(), c_void_p), arr(xd, lambda t: t.stride(0), ctypes.c_int64),
                                      arr(xd, lambda t: t.size(1), ctypes.c_int32), len(xd),
                                      arr(gd, lambda t: t.data_ptr(), c_void_p), arr(gd, lambda t: t.stride(0), ctypes.c_int64),
                                      arr(gd, lambda t: t.size(1), ctypes.c_int32), len(gd), n, code, ptr(out), ptr(ws),
                                      need.value, stream_ptr()), "pygsd_tall_gram")
        return out if dtype == torch.float32 else out.to(dtype)
    x = xs[0] if len(xs) == 1 else torch.cat(xs, dim=1)
    g = gs[0] if len(gs) == 1 else torch.cat(gs, dim=1)
    if _TALL_KERNELS and x.is_cuda and dtype == torch.float32 and g.dtype == torch.float32:
        return gemm(x.t(), g)                       # generic HIP GEMM, reduction over the rows split over the blocks
    if _TALL_KERNELS and x.is_cuda and dtype == torch.bfloat16 and g.dtype == torch.bfloat16:
        return gemm_bf16(x.t(), g, out_dtype=torch.float32).to(dtype)      # (fp32 sums, rounded once -- as the fused path)
    if x.is_cuda:
        _cabi.note_library_route("tall_gram", f"{tuple(x.shape)}^T {tuple(g.shape)} {dtype}")
    slab = _TallLinear.SLAB
    s = n // slab
    if s < 8:
        return x.t() @ g
    x, g = x.contiguous(), g.contiguous()   # column slices of a wider matrix: one copy beats the skinny GEMM
    head = s * slab
    gw = torch.bmm(x[:head].view(s, slab, -1).transpose(1, 2), g[:head].view(s, slab, -1)).sum(0)
    if head < n:
        gw = gw + x[head:].t() @ g[head:]
    return gw


def tall_linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """x [N, F_in] @ weight [F_in, F_out] (+ bias) with a split-K weight gradient."""
    return _TallLinear.apply(x, weight, bias)
