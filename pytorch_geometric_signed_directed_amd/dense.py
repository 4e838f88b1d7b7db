"""Host wrappers of the fused MFMA dense stage (csrc/dense.hip; include/pygsd_hip.h
pygsd_magnetic_dense_*) and the single autograd node of a whole MagNetConv / MSConv layer."""
import ctypes
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


def dense_fwd_raw(a: List[Tensor], b: List[Tensor], weight: Tensor, bias: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """out_real = sum_k (a_k - b_k) W_k + bias ; out_imag = sum_k (a_k + b_k) W_k + bias."""
    k1, f_in, f_out = weight.shape
    for t in list(a) + list(b) + [weight]:
        if t.dtype != torch.float32:
            raise TypeError(f"the fused dense stage computes in float32; got {t.dtype}")
    a = [t.contiguous() for t in a]
    b = [t.contiguous() for t in b]
    n = a[0].size(0)
    w = weight.detach().contiguous()
    out_r = torch.empty((n, f_out), dtype=torch.float32, device=w.device)
    out_i = torch.empty_like(out_r)
    bias_c = None if bias is None else bias.detach().contiguous()
    with torch.cuda.device(w.device):
        check(_cabi.lib().pygsd_magnetic_dense_fwd_f32(_ptr_array(a), _ptr_array(b), k1, ptr(w), ptr(bias_c),
                                                       ptr(out_r), ptr(out_i), n, f_in, f_out, stream_ptr()),
              "pygsd_magnetic_dense_fwd_f32")
    return out_r, out_i


def dense_bwd_raw(a: List[Tensor], b: List[Tensor], weight: Tensor, g_r: Tensor, g_i: Tensor,
                  rows: Optional[int] = None):
    """-> (da list, db list, dW [k1, f_in, f_out], dbias [f_out]).
    rows: only the first `rows` rows enter (the sharded layer's real rows; the pad rows behind them are not part of
    the graph): dW / dbias sum over those rows, da / db keep the full height with zero rows behind."""
    k1, f_in, f_out = weight.shape
    a = [t.contiguous() for t in a]
    b = [t.contiguous() for t in b]
    if g_r.size(0) > 1 and g_r.stride(0) == 0 and g_i.stride(0) == 0:
        # one row broadcast to every node (the gradient of a loss that sums over the nodes arrives as an expanded
        # tensor): hand the kernel that row with a zero row stride instead of materialising two [N, F] copies
        g_r, g_i, ldg = g_r[:1].contiguous(), g_i[:1].contiguous(), 0
    else:
        g_r, g_i, ldg = g_r.contiguous(), g_i.contiguous(), f_out
    n_full = a[0].size(0)
    n = n_full if rows is None else int(rows)
    dev = weight.device
    w = weight.detach().contiguous()
    da = [torch.empty((n_full, f_in), dtype=torch.float32, device=dev) for _ in range(k1)]
    db = [torch.empty((n_full, f_in), dtype=torch.float32, device=dev) for _ in range(k1)]
    if n < n_full:
        for t in da + db:
            t[n:].zero_()
    if n == 0:                                     # a shard without real rows contributes nothing
        return da, db, torch.zeros((k1, f_in, f_out), dtype=torch.float32, device=dev), \
            torch.zeros(f_out, dtype=torch.float32, device=dev)
    dw = torch.empty((k1, f_in, f_out), dtype=torch.float32, device=dev)
    dbias = torch.empty(f_out, dtype=torch.float32, device=dev)
    lib = _cabi.lib()
    with torch.cuda.device(dev):
        need = ctypes.c_size_t(0)
        check(lib.pygsd_magnetic_dense_bwd_workspace(n, f_in, f_out, k1, ctypes.byref(need)),
              "pygsd_magnetic_dense_bwd_workspace")
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        check(lib.pygsd_magnetic_dense_bwd_f32(_ptr_array(a), _ptr_array(b), k1, ptr(w), ptr(g_r), ptr(g_i), ldg,
                                               _ptr_array(da), _ptr_array(db), ptr(dw), ptr(dbias), n, f_in,
                                               f_out, ptr(ws), need.value, stream_ptr()),
              "pygsd_magnetic_dense_bwd_f32")
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
        ta, tb = [x_real.contiguous()], [x_imag.contiguous()]
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


class _TallLinear(torch.autograd.Function):
    """y = x @ W (+ b) for a tall x [N, F_in] (N ~ 10^5..10^6, F ~ 10^1..10^2) with library GEMMs.
    Forward and dX are ordinary GEMMs; the weight gradient dW = x^T g has a reduction dimension of N and
    only F_in x F_out outputs, which rocBLAS runs on a handful of CUs (measured 1.8 ms at N = 10^6,
    F = 64).  Here it is a batched split-K: N is cut into 4096-row slabs, one GEMM per slab (bmm), and
    the [S, F_in, F_out] partials are summed -- every CU gets work, ~10x faster."""
    SLAB = 4096

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        y = torch.matmul(x, weight)
        return y if bias is None else y + bias

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.matmul(g, weight.t())
        if ctx.needs_input_grad[1]:
            gw = tall_gram(x, g)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return gx, gw, gb


def tall_gram(x: Tensor, g: Tensor) -> Tensor:
    """x^T g for tall x [N, F_in], g [N, F_out] (N ~ 10^5..10^6): batched split-K over 4096-row slabs (see _TallLinear)."""
    n, slab = x.size(0), _TallLinear.SLAB
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
