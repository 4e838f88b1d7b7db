"""SGCNConv -- drop-in for torch_geometric_signed_directed/nn/signed/SGCNConv.py:16: mean
aggregation over positive / negative incoming edges (value-less segment-mean HIP kernel), concat
with the node's own features, Linear."""
from typing import Tuple, Union

import torch
import torch.nn.functional as F
from torch import Tensor

from ... import _cabi
from ...message_passing import MessagePassing
from ...dense import tall_linear
from ...sparse import GLOBAL_PATTERNS, spmm


class SGCNConv(MessagePassing):
    edge_weight_arg = None
    _fused_message = True

    def __init__(self, in_dim: int, out_dim: int, first_aggr: bool, bias: bool = True,
                 norm_emb: bool = False, **kwargs):
        kwargs.setdefault('aggr', 'mean')
        super().__init__(**kwargs)
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.first_aggr = first_aggr
        self.norm_emb = norm_emb
        k = 2 if first_aggr else 3
        self.lin_b = torch.nn.Linear(k * in_dim, out_dim, bias)
        self.lin_u = torch.nn.Linear(k * in_dim, out_dim, bias)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin_b.reset_parameters()
        self.lin_u.reset_parameters()

    def _mean_in(self, x_src: Tensor, n_dst: int, edge_index: Tensor) -> Tensor:
        pat = GLOBAL_PATTERNS.get(edge_index, x_src.size(0), n_dst, self.flow)
        return spmm(pat, x_src, None, reduce=self.aggr)

    def _branch(self, lin, aggregated, own, n):
        """lin(cat([mean_in(x_1), ..., own])).  The Linear is applied block-wise; when it narrows the
        features (in_dim > out_dim) each block is multiplied BEFORE its mean aggregation --
        mean_in(x) W = mean_in(x W) -- so the segment-mean SpMM runs at width out_dim
        (reference order: aggregate, concatenate, then Linear; SGCNConv.py:101-119)."""
        f = self.in_dim
        w = lin.weight                                  # [out_dim, (len(aggregated) + 1) * in_dim]
        if self.in_dim > self.out_dim:
            out = tall_linear(own, w[:, len(aggregated) * f:].t(), lin.bias)
            for k, (feat, ei) in enumerate(aggregated):
                out = out + self._mean_in(tall_linear(feat, w[:, k * f:(k + 1) * f].t()), n, ei)
            return out
        parts = [self._mean_in(feat, n, ei) for feat, ei in aggregated] + [own]
        return tall_linear(torch.cat(parts, dim=-1), w.t(), lin.bias)

    def _fused(self, x: Tensor, pos_edge_index: Tensor, neg_edge_index: Tensor) -> Tensor:
        """Both branches from ONE GEMM when the Linear does not widen (in_dim >= out_dim): every block of
        lin_b / lin_u becomes a column block of one [F_x, 4 or 6 * out_dim] matrix (zero blocks where a block
        reads the other half of x), the biases ride on the own-feature blocks, and each mean aggregation adds
        its result onto the previous block through the SpMM's beta * Z epilogue -- no concatenation, no
        element-wise adds, one weight-gradient GEMM."""
        f, o, n = self.in_dim, self.out_dim, x.size(0)
        wb, wu = self.lin_b.weight, self.lin_u.weight
        zeros = wb.new_zeros(f, o)
        if self.first_aggr:             # columns: own_b | agg_b(pos) | own_u | agg_u(neg)
            w_big = torch.cat([wb[:, f:].t(), wb[:, :f].t(), wu[:, f:].t(), wu[:, :f].t()], dim=1)
            own = (0, 2)
        else:                           # x = [lo | hi]; columns: own_b | pos_b | neg_b | own_u | pos_u | neg_u
            top = torch.cat([wb[:, 2 * f:].t(), wb[:, :f].t(), zeros, zeros, zeros, wu[:, f:2 * f].t()], dim=1)
            bot = torch.cat([zeros, zeros, wb[:, f:2 * f].t(), wu[:, 2 * f:].t(), wu[:, :f].t(), zeros], dim=1)
            w_big = torch.cat([top, bot], dim=0)
            own = (0, 3)
        bias = None
        if self.lin_b.bias is not None:
            z = self.lin_b.bias.new_zeros(o)
            blocks = [z] * (w_big.size(1) // o)
            blocks[own[0]], blocks[own[1]] = self.lin_b.bias, self.lin_u.bias
            bias = torch.cat(blocks)
        y = tall_linear(x, w_big, bias).split(o, dim=1)
        pos = GLOBAL_PATTERNS.get(pos_edge_index, n, n, self.flow)
        neg = GLOBAL_PATTERNS.get(neg_edge_index, n, n, self.flow)
        if self.first_aggr:
            out_b = spmm(pos, y[1], None, z=y[0], beta=1.0, reduce=self.aggr)
            out_u = spmm(neg, y[3], None, z=y[2], beta=1.0, reduce=self.aggr)
        else:
            out_b = spmm(neg, y[2], None, z=spmm(pos, y[1], None, z=y[0], beta=1.0, reduce=self.aggr), beta=1.0,
                         reduce=self.aggr)
            out_u = spmm(neg, y[5], None, z=spmm(pos, y[4], None, z=y[3], beta=1.0, reduce=self.aggr), beta=1.0,
                         reduce=self.aggr)
        return torch.cat([out_b, out_u], dim=-1)

    def forward(self, x: Union[Tensor, Tuple[Tensor, Tensor]], pos_edge_index: Tensor,
                neg_edge_index: Tensor) -> Tensor:
        if (isinstance(x, Tensor) and x.dim() == 2 and self.in_dim >= self.out_dim
                and isinstance(pos_edge_index, Tensor) and isinstance(neg_edge_index, Tensor)):
            _cabi.require_gpu(x, pos_edge_index, neg_edge_index)
            out = self._fused(x, pos_edge_index, neg_edge_index)
            return F.normalize(out, p=2, dim=-1) if self.norm_emb else out
        if isinstance(x, Tensor):
            x = (x, x)
        if not isinstance(pos_edge_index, Tensor) or not isinstance(neg_edge_index, Tensor):
            raise NotImplementedError("SGCNConv: only the edge_index (Tensor) path exists on the HIP stack")
        _cabi.require_gpu(x[0], x[1], pos_edge_index, neg_edge_index)
        n = x[1].size(0)
        if self.first_aggr:
            out_b = self._branch(self.lin_b, [(x[0], pos_edge_index)], x[1], n)
            out_u = self._branch(self.lin_u, [(x[0], neg_edge_index)], x[1], n)
        else:
            f = self.in_dim
            lo, hi = x[0][..., :f], x[0][..., f:]   # column slices: passed by row stride, no copy
            out_b = self._branch(self.lin_b, [(lo, pos_edge_index), (hi, neg_edge_index)], x[1][..., :f], n)
            out_u = self._branch(self.lin_u, [(hi, pos_edge_index), (lo, neg_edge_index)], x[1][..., f:], n)
        out = torch.cat([out_b, out_u], dim=-1)
        if self.norm_emb:
            out = F.normalize(out, p=2, dim=-1)
        return out

    def message(self, x_j: Tensor) -> Tensor:
        return x_j

    def __repr__(self) -> str:
        return (f'{self.__class__.__name__}({self.in_dim}, '
                f'{self.out_dim}, first_aggr={self.first_aggr})')
