"""GATConv + SDRLayer -- the attention aggregate under SDGNN / SiGAT.

The reference builds its signed-directed-relationship layer (nn/signed/SDGNN.py:13-64) and SiGAT's
aggregators (nn/signed/SiGAT.py:59-64) from `torch_geometric.nn.GATConv` with all defaults.  PyG is not part
of this stack, so GATConv is restated here (current PyG layout: one bias-free `lin`, `att_src`, `att_dst`
[1, heads, out], `bias`; state_dict keys att_src, att_dst, bias, lin.weight) over the HIP kernels:
segment-softmax coefficients (pygsd_gat_alpha_csr_f32), weighted aggregate (pygsd_spmm_csr_f32), and a
fused SDDMM + softmax/leaky-relu backward (pygsd_gat_alpha_bwd_csr_f32)."""
import math
from typing import List

import torch
import torch.nn as nn

from ... import _cabi, memo
from ..._cabi import check, ptr, stream_ptr
from ...dense import tall_linear
from ...sparse import (GLOBAL_PATTERNS, Pattern, _rows, _spmm_raw, gather_values, segment_long_rows_arg,
                       segment_sum_raw)


def _row_sum(csr, w_coo):
    """Per-row sums of COO-ordered values (gradient reductions: no reference summation order to keep), hub rows
    through the segment-parallel path."""
    return segment_sum_raw(csr.rowptr, csr.perm, w_coo, csr.n_rows, csr)


class _GatAggregate(torch.autograd.Function):
    """out_i = sum_j softmax_i(leaky_relu(a_src[j] + a_dst[i])) h_j over the pattern's incoming edges."""

    @staticmethod
    def forward(ctx, h, a_src, a_dst, pat: Pattern, slope: float):
        _cabi.require_gpu(h, a_src, a_dst)
        h, _ = _rows(h.float())
        a_src, a_dst = a_src.float().contiguous(), a_dst.float().contiguous()
        csr = pat.fwd
        alpha = torch.empty(csr.nnz, dtype=torch.float32, device=h.device)
        if csr.nnz:
            with _cabi.on_device(h.device):
                hubs, keep = segment_long_rows_arg(csr)
                check(_cabi.lib().pygsd_gat_alpha_csr_f32(ptr(csr.rowptr), ptr(csr.col), ptr(a_src), ptr(a_dst),
                                                          csr.n_rows, float(slope), ptr(alpha), hubs, stream_ptr()),
                      "pygsd_gat_alpha_csr_f32")
                del keep
        out = _spmm_raw(csr, alpha if csr.nnz else None, h, None, 1.0, 0.0, False)
        ctx.pat, ctx.slope = pat, slope
        ctx.save_for_backward(h, a_src, a_dst, alpha, out)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        h, a_src, a_dst, alpha, out = ctx.saved_tensors
        pat = ctx.pat
        fwd, bwd = pat.fwd, pat.bwd
        g, ldg = _rows(g.contiguous())
        hh, ldh = _rows(h)
        oo, ldo = _rows(out)
        ds = torch.empty(fwd.nnz, dtype=torch.float32, device=h.device)
        f = h.size(1)
        if fwd.nnz and f % 4 == 0 and f <= 256 and ldg % 4 == 0 and ldh % 4 == 0 and ldo % 4 == 0 and \
                all(t.data_ptr() % 16 == 0 for t in (g, hh, oo)):
            # vectorised path: ds in by-target slot order, d a_dst from the same pass
            da_dst = torch.empty(fwd.n_rows, dtype=torch.float32, device=h.device)
            with _cabi.on_device(h.device):
                hubs, keep = segment_long_rows_arg(fwd)
                check(_cabi.lib().pygsd_gat_alpha_bwd_csr_v2_f32(ptr(fwd.rowptr), ptr(fwd.col), ptr(a_src), ptr(a_dst),
                                                                 float(ctx.slope), ptr(alpha), ptr(hh), ldh, ptr(g),
                                                                 ldg, ptr(oo), ldo, fwd.n_rows, f, ptr(ds),
                                                                 ptr(da_dst), hubs, stream_ptr()),
                      "pygsd_gat_alpha_bwd_csr_v2_f32")
                del keep
            m = pat.bwd_to_fwd
            gh = _spmm_raw(bwd, gather_values(alpha, m), g, None, 1.0, 0.0, False)
            return gh, segment_sum_raw(bwd.rowptr, m, ds, bwd.n_rows, bwd), da_dst, None, None
        a_coo = torch.empty_like(ds)
        if fwd.nnz:
            with _cabi.on_device(h.device):
                hubs, keep = segment_long_rows_arg(fwd)
                check(_cabi.lib().pygsd_gat_alpha_bwd_csr_f32(ptr(fwd.rowptr), ptr(fwd.col), ptr(fwd.perm), ptr(a_src),
                                                              ptr(a_dst), float(ctx.slope), ptr(alpha), ptr(hh), ldh,
                                                              ptr(g), ldg, ptr(oo), ldo, fwd.n_rows, h.size(1),
                                                              ptr(ds), ptr(a_coo), hubs, stream_ptr()),
                      "pygsd_gat_alpha_bwd_csr_f32")
                del keep
        gh = _spmm_raw(bwd, gather_values(a_coo, bwd.perm) if bwd.nnz else None, g, None, 1.0, 0.0, False)
        return gh, _row_sum(bwd, ds), _row_sum(fwd, ds), None, None


class GATConv(nn.Module):
    r"""Graph attention convolution with torch_geometric.nn.GATConv's default behaviour (heads=1,
    concat=True, negative_slope=0.2, add_self_loops=True, bias=True).  Dropout on the attention
    coefficients is not supported on the fused path (the reference uses the default 0)."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = True,
                 negative_slope: float = 0.2, dropout: float = 0.0, add_self_loops: bool = True,
                 bias: bool = True, **kwargs):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError("GATConv: attention dropout is not on the HIP path")
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.dropout = concat, negative_slope, dropout
        self.add_self_loops = add_self_loops
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(heads * out_channels if concat else out_channels))
        else:
            self.register_parameter('bias', None)
        self._loops_memo = memo.TensorMemo(1)
        self.reset_parameters()

    def reset_parameters(self):
        for t in (self.lin.weight, self.att_src, self.att_dst):
            a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
            t.data.uniform_(-a, a)
        if self.bias is not None:
            self.bias.data.fill_(0)

    def _with_self_loops(self, edge_index, n):
        """remove_self_loops then add_self_loops (pure function of the edge list; memoised on the tensor)."""
        hit = self._loops_memo.get((edge_index,), n)
        if hit is not None:
            return hit
        from ...utils._norm import add_remaining_self_loops
        out, _ = add_remaining_self_loops(edge_index, None, 1.0, n, with_weights=False)
        memo.own(out)                      # derived here, held by nobody else: the pattern lookup need not re-check its content
        return self._loops_memo.put((edge_index,), n, out)

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
        _cabi.require_gpu(x, edge_index)
        n, hds, c = x.size(0), self.heads, self.out_channels
        # h = lin(x) and the attention projections <h_k, att_src_k>, <h_k, att_dst_k> from ONE GEMM: the
        # projections are 2 * heads extra output columns W_k^T att_k of the linear map
        wt = self.lin.weight.t()                                        # [in, heads * c]
        w3 = wt.reshape(wt.size(0), hds, c)
        y = tall_linear(x, torch.cat([wt, (w3 * self.att_src).sum(-1), (w3 * self.att_dst).sum(-1)], dim=1))
        h = y[:, :hds * c].contiguous().view(n, hds, c)   # 16-byte aligned, densely packed rows for the gathers
        a_src, a_dst = y[:, hds * c:hds * c + hds], y[:, hds * c + hds:]
        if self.add_self_loops:
            edge_index = self._with_self_loops(edge_index, n)
        pat = GLOBAL_PATTERNS.get(edge_index, n, n, "source_to_target")
        outs = [_GatAggregate.apply(h[:, k], a_src[:, k], a_dst[:, k], pat, self.negative_slope)
                for k in range(hds)]
        out = torch.cat(outs, dim=1) if self.concat else torch.stack(outs, dim=1).mean(dim=1)
        return out if self.bias is None else out + self.bias

    def __repr__(self):
        return f'{self.__class__.__name__}({self.in_channels}, {self.out_channels}, heads={self.heads})'


class SDRLayer(nn.Module):
    r"""The signed directed relationship layer of SDGNN (reference nn/signed/SDGNN.py:13-64): one GATConv
    per motif edge list, concatenated with the input, then Linear-Tanh-Linear."""

    def __init__(self, in_dim: int = 20, out_dim: int = 20, edge_lists: List[torch.Tensor] = [], **kwargs):
        super().__init__(**kwargs)
        self.edge_lists = edge_lists
        self.aggs = []
        for i in range(len(edge_lists)):
            self.aggs.append(GATConv(in_dim, out_dim))
            self.add_module('agg_{}'.format(i), self.aggs[-1])
        self.mlp_layer = nn.Sequential(nn.Linear(in_dim * (len(edge_lists) + 1), out_dim), nn.Tanh(),
                                       nn.Linear(out_dim, out_dim))

    def reset_parameters(self):
        def init_weights(m):
            if type(m) == nn.Linear:
                torch.nn.init.kaiming_normal_(m.weight)
        self.mlp_layer.apply(init_weights)
        for agg in self.aggs:
            agg.reset_parameters()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        neigh = [agg(x, edges) for edges, agg in zip(self.edge_lists, self.aggs)]
        return self.mlp_layer(torch.cat([x] + neigh, 1))
