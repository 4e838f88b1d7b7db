"""Differentiable per-segment (= per CSR row) reductions over values that are already in CSR order:
`segment_softmax` (pygsd_segment_softmax_csr_f32 / _bwd_) and `segment_sum` (pygsd_csr_row_sum_f32).
The building blocks of attention layers whose per-edge logits are not the GAT form a_src[j] + a_dst[i]
(SNEAConv, reference nn/signed/SNEAConv.py:135-146)."""
import torch

from . import _cabi
from ._cabi import check, ptr, stream_ptr
from .sparse import CSR, segment_long_rows_arg, segment_sum_raw

Tensor = torch.Tensor


def row_ids(csr: CSR) -> Tensor:
    """int64 [nnz]: the row of every CSR slot (for torch-side gathers of per-row quantities)."""
    deg = (csr.rowptr[1:] - csr.rowptr[:-1]).long()
    return torch.repeat_interleave(torch.arange(csr.n_rows, device=deg.device), deg, output_size=csr.nnz)


class _SegmentSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: Tensor, csr: CSR):
        _cabi.require_gpu(logits)
        logits = logits.float().contiguous()
        if logits.numel() != csr.nnz:
            raise ValueError(f"segment_softmax: {logits.numel()} logits for {csr.nnz} CSR entries")
        alpha = torch.empty_like(logits)
        if csr.nnz:
            with _cabi.on_device(logits.device):
                hubs, keep = segment_long_rows_arg(csr)
                check(_cabi.lib().pygsd_segment_softmax_csr_f32(ptr(csr.rowptr), ptr(logits), csr.n_rows, ptr(alpha),
                                                                hubs, stream_ptr()), "pygsd_segment_softmax_csr_f32")
                del keep
        ctx.csr = csr
        ctx.save_for_backward(alpha)
        return alpha

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (alpha,) = ctx.saved_tensors
        csr = ctx.csr
        g = g.float().contiguous()
        out = torch.empty_like(alpha)
        if csr.nnz:
            with _cabi.on_device(alpha.device):
                hubs, keep = segment_long_rows_arg(csr)
                check(_cabi.lib().pygsd_segment_softmax_bwd_csr_f32(ptr(csr.rowptr), ptr(alpha), ptr(g), csr.n_rows,
                                                                    ptr(out), hubs, stream_ptr()),
                      "pygsd_segment_softmax_bwd_csr_f32")
                del keep
        return out, None


class _SegmentSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vals: Tensor, csr: CSR, rows: Tensor):
        _cabi.require_gpu(vals)
        vals = vals.float().contiguous()
        if vals.numel() != csr.nnz:
            raise ValueError(f"segment_sum: {vals.numel()} values for {csr.nnz} CSR entries")
        if csr.nnz and csr.n_rows and csr.hubs() is not None:
            # hub rows: the sequential reference-order row sum below would take one thread through 10^5..10^6
            # entries; the team / segment-parallel sum handles them (fixed, different summation order)
            out = segment_sum_raw(csr.rowptr, None, vals, csr.n_rows, csr)
            ctx.save_for_backward(rows)
            return out
        out = torch.zeros(csr.n_rows, dtype=torch.float32, device=vals.device)
        if csr.nnz and csr.n_rows:
            with _cabi.on_device(vals.device):
                check(_cabi.lib().pygsd_csr_row_sum_f32(ptr(csr.rowptr), None, ptr(vals), csr.n_rows, ptr(out),
                                                        stream_ptr()), "pygsd_csr_row_sum_f32")
        ctx.save_for_backward(rows)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (rows,) = ctx.saved_tensors
        return g[rows], None, None


def segment_softmax(csr: CSR, logits: Tensor) -> Tensor:
    """softmax of `logits` (CSR order) within each CSR row: exp(l - max) / (sum + 1e-16)."""
    return _SegmentSoftmax.apply(logits, csr)


def segment_sum(csr: CSR, vals: Tensor, rows: Tensor = None) -> Tensor:
    """[n_rows] sums of `vals` (CSR order) per CSR row; `rows` = row_ids(csr) if the caller has it cached."""
    return _SegmentSum.apply(vals, csr, row_ids(csr) if rows is None else rows)
