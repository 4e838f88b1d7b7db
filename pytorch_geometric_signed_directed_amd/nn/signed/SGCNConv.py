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

    def forward(self, x: Union[Tensor, Tuple[Tensor, Tensor]], pos_edge_index: Tensor,
                neg_edge_index: Tensor) -> Tensor:
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
