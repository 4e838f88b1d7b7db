"""SNEAConv -- the signed attention layer of SNEA (reference nn/signed/SNEAConv.py:9-150).

What the reference's propagate/message computes (verified against it, tests/golden/snea_*.npz):
  * per edge (j -> i) of type p (0: positive or self loop, 1: negative) a logit
    tanh(alpha_func([x_p[j], x_p[i]])) with x_0 = x1, x_1 = x2 -- i.e. tanh(<x_p[j], a_src> + <x_p[i], a_dst> + b);
  * alpha = softmax of the logits over ALL edges into i (both types together);
  * the message is `x_i * alpha` -- the TARGET's own row (x1_i or x2_i by edge type), not the neighbour's --
    so out_i = x1_i * (sum of alpha over type-0 edges) + x2_i * (sum over type-1 edges).
  * self loops: removed, then re-added for nodes 0 .. max id appearing in the remaining edges (add_self_loops
    with num_nodes=None), so trailing nodes without edges get NO loop and a zero output row.
Device path: the per-node projections s_t = <x_t, a_src>, d_t = <x_t, a_dst> and the final row scaling are
N x F torch ops; everything per EDGE (logits, softmax, per-type shares, and their backward) runs in the fused
HIP kernels pygsd_snea_alpha_csr_f32 / pygsd_snea_alpha_bwd_csr_f32, the by-source sums of the backward in
pygsd_csr_row_sum_f32."""
from typing import Optional

import torch
import torch.nn as nn

from ... import _cabi
from ..._cabi import check, ptr, stream_ptr
from ...dense import tall_linear
from ...memo import TensorMemo
from ...sparse import Pattern, segment_long_rows_arg, segment_sum_raw


def _loop_free_plus_loops(edge_index: torch.Tensor) -> torch.Tensor:
    e = edge_index[:, edge_index[0] != edge_index[1]]
    n = int(e.max()) + 1 if e.numel() > 0 else 0
    loops = torch.arange(n, dtype=e.dtype, device=e.device)
    return torch.cat([e, torch.stack([loops, loops])], dim=1)


class _Graph:
    """One combined edge list grouped by target (forward) and by source (backward of the s_t projections),
    the per-slot edge type in by-target order, and for every by-source slot its by-target slot."""

    def __init__(self, edge_index: torch.Tensor, edge_p: Optional[torch.Tensor], n: int):
        pat = Pattern(edge_index, n, n)
        self.fwd, self.bwd = pat.fwd, pat.bwd
        self.etype = None if edge_p is None else edge_p[self.fwd.perm.long()].to(torch.uint8).contiguous()
        self.bwd_to_fwd = pat.bwd_to_fwd


class _SneaShares(torch.autograd.Function):
    """(s0, s1, d0, d1, bias) -> (share0, share1): per-row sums of the attention coefficients per edge type."""

    @staticmethod
    def forward(ctx, s0, s1, d0, d1, bias, g: _Graph):
        typed = g.etype is not None
        f32 = lambda t: None if t is None else t.detach().float().contiguous()  # noqa: E731
        s0, s1, d0, d1, bias = f32(s0), f32(s1) if typed else None, f32(d0), f32(d1) if typed else None, f32(bias)
        n = g.fwd.n_rows
        alpha = torch.empty(g.fwd.nnz, dtype=torch.float32, device=s0.device)
        share0 = torch.zeros(n, dtype=torch.float32, device=s0.device)
        share1 = torch.zeros(n, dtype=torch.float32, device=s0.device) if typed else None
        if g.fwd.nnz:
            with _cabi.on_device(s0.device):
                hubs, keep = segment_long_rows_arg(g.fwd)
                check(_cabi.lib().pygsd_snea_alpha_csr_f32(ptr(g.fwd.rowptr), ptr(g.fwd.col), ptr(g.etype), ptr(s0),
                                                           ptr(s1), ptr(d0), ptr(d1), ptr(bias), n, ptr(alpha),
                                                           ptr(share0), ptr(share1), hubs, stream_ptr()),
                      "pygsd_snea_alpha_csr_f32")
                del keep
        ctx.g, ctx.typed = g, typed
        ctx.save_for_backward(s0, s1, d0, d1, bias, alpha)
        if typed:
            return share0, share1
        ctx.mark_non_differentiable(z := torch.zeros_like(share0))
        return share0, z

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g0, g1):
        s0, s1, d0, d1, bias, alpha = ctx.saved_tensors
        g, typed = ctx.g, ctx.typed
        n, nnz = g.fwd.n_rows, g.fwd.nnz
        dev = alpha.device
        g0 = g0.float().contiguous()
        g1 = g1.float().contiguous() if typed else None
        dpre0 = torch.empty(nnz, dtype=torch.float32, device=dev)
        dpre1 = torch.empty(nnz, dtype=torch.float32, device=dev) if typed else None
        dd0 = torch.zeros(n, dtype=torch.float32, device=dev)
        dd1 = torch.zeros(n, dtype=torch.float32, device=dev) if typed else None
        if nnz:
            with _cabi.on_device(dev):
                hubs, keep = segment_long_rows_arg(g.fwd)
                check(_cabi.lib().pygsd_snea_alpha_bwd_csr_f32(ptr(g.fwd.rowptr), ptr(g.fwd.col), ptr(g.etype), ptr(s0),
                                                               ptr(s1), ptr(d0), ptr(d1), ptr(bias), ptr(alpha),
                                                               ptr(g0), ptr(g1), n, ptr(dpre0), ptr(dpre1), ptr(dd0),
                                                               ptr(dd1), hubs, stream_ptr()),
                      "pygsd_snea_alpha_bwd_csr_f32")
                del keep
        ds0 = segment_sum_raw(g.bwd.rowptr, g.bwd_to_fwd, dpre0, n, g.bwd)
        ds1 = segment_sum_raw(g.bwd.rowptr, g.bwd_to_fwd, dpre1, n, g.bwd) if typed else None
        dbias = (dd0.sum() + dd1.sum() if typed else dd0.sum()).reshape(1)
        return ds0, ds1, dd0, dd1, dbias, None


class SNEAConv(nn.Module):
    def __init__(self, in_dim: int, out_dim: int, first_aggr: bool, bias: bool = True, norm_emb: bool = True,
                 add_self_loops=True, **kwargs):
        super().__init__()
        self.in_dim, self.out_dim, self.first_aggr = in_dim, out_dim, first_aggr
        self.add_self_loops, self.norm_emb = add_self_loops, norm_emb        # stored, unused (as in the reference)
        self.lin_b = nn.Linear(in_dim, out_dim, bias)
        self.lin_u = nn.Linear(in_dim, out_dim, bias)
        self.alpha_u = nn.Linear(out_dim * 2, 1)
        self.alpha_b = nn.Linear(out_dim * 2, 1)
        self._memo = TensorMemo(1)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin_b.reset_parameters()
        self.lin_u.reset_parameters()
        nn.init.xavier_normal_(self.alpha_b.weight)
        nn.init.xavier_normal_(self.alpha_u.weight)

    def _graphs(self, pos, neg, n):
        hit = self._memo.get((pos, neg), n)
        if hit is not None:
            return hit
        if self.first_aggr:
            out = (_Graph(_loop_free_plus_loops(pos), None, n), _Graph(_loop_free_plus_loops(neg), None, n))
        else:
            e1 = _loop_free_plus_loops(pos)
            e2 = neg[:, neg[0] != neg[1]]
            flags = torch.cat([torch.zeros(e1.size(1), dtype=torch.bool, device=pos.device),
                               torch.ones(e2.size(1), dtype=torch.bool, device=pos.device)])
            out = (_Graph(torch.cat([e1, e2], dim=1), flags, n),)
        return self._memo.put((pos, neg), n, out)

    def _project(self, lin, alpha_func, x):
        """(lin(x), <lin(x), a_src>, <lin(x), a_dst>) from ONE GEMM: the two attention projections are two
        extra output columns W^T a of the linear map (rocBLAS' gemv on a 5e5 x 32 operand takes 4.5 ms; as
        GEMM columns they are free), weight gradient through the split-K tall_linear."""
        o = self.out_dim
        a = alpha_func.weight[0].view(2, o).t()                     # [out, 2] = (a_src | a_dst)
        wt = lin.weight.t()
        y = tall_linear(x, torch.cat([wt, wt @ a], dim=1),
                        None if lin.bias is None else torch.cat([lin.bias, lin.bias @ a]))
        return y[:, :o], y[:, o], y[:, o + 1]

    def _aggregate(self, g: _Graph, lin, alpha_func, h1, h2=None):
        x1, s0, d0 = self._project(lin, alpha_func, h1)
        if g.etype is None:
            share0, _ = _SneaShares.apply(s0, None, d0, None, alpha_func.bias, g)
            return x1 * share0.unsqueeze(1)
        x2, s1, d1 = self._project(lin, alpha_func, h2)
        share0, share1 = _SneaShares.apply(s0, s1, d0, d1, alpha_func.bias, g)
        return x1 * share0.unsqueeze(1) + x2 * share1.unsqueeze(1)

    def forward(self, x: torch.Tensor, pos_edge_index: torch.Tensor, neg_edge_index: torch.Tensor) -> torch.Tensor:
        _cabi.require_gpu(x, pos_edge_index, neg_edge_index)
        graphs = self._graphs(pos_edge_index, neg_edge_index, x.size(0))
        if self.first_aggr:
            out_b = self._aggregate(graphs[0], self.lin_b, self.alpha_b, x)
            out_u = self._aggregate(graphs[1], self.lin_u, self.alpha_u, x)
        else:
            h_b, h_u = x[..., :self.in_dim], x[..., self.in_dim:]
            out_b = self._aggregate(graphs[0], self.lin_b, self.alpha_b, h_b, h_u)
            out_u = self._aggregate(graphs[0], self.lin_u, self.alpha_u, h_u, h_b)
        return torch.cat([out_b, out_u], dim=-1)

    def __repr__(self) -> str:
        return f'{self.__class__.__name__}({self.in_dim}, {self.out_dim}, first_aggr={self.first_aggr})'
