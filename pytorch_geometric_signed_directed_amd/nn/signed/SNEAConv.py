"""SNEAConv -- the signed attention layer of SNEA (reference nn/signed/SNEAConv.py:9-150).

What the reference's propagate/message computes (verified against it, tests/golden/snea_*.npz):
  * per edge (j -> i) of type p (0: positive or self loop, 1: negative) a logit
    tanh(alpha_func([x_p[j], x_p[i]])) with x_0 = x1, x_1 = x2 -- i.e. tanh(<x_p[j], a_src> + <x_p[i], a_dst> + b);
  * alpha = softmax of the logits over ALL edges into i (both types together);
  * the message is `x_i * alpha` -- the TARGET's own row (x1_i or x2_i by edge type), not the neighbour's --
    so out_i = x1_i * (sum of alpha over type-0 edges) + x2_i * (sum over type-1 edges).
  * self loops: removed, then re-added for nodes 0 .. max id appearing in the remaining edges (add_self_loops
    with num_nodes=None), so trailing nodes without edges get NO loop and a zero output row.
Device path: per-edge scalars through torch gathers, softmax and the per-type sums through the HIP segment
kernels (segment.py)."""
from typing import Optional

import torch
import torch.nn as nn

from ... import _cabi
from ...segment import row_ids, segment_softmax, segment_sum
from ...sparse import Pattern


def _loop_free_plus_loops(edge_index: torch.Tensor) -> torch.Tensor:
    e = edge_index[:, edge_index[0] != edge_index[1]]
    n = int(e.max()) + 1 if e.numel() > 0 else 0
    loops = torch.arange(n, dtype=e.dtype, device=e.device)
    return torch.cat([e, torch.stack([loops, loops])], dim=1)


class _Graph:
    """CSR by target of one combined edge list + per-slot source ids, row ids and type flags."""

    def __init__(self, edge_index: torch.Tensor, edge_p: Optional[torch.Tensor], n: int):
        self.csr = Pattern(edge_index, n, n).fwd
        self.src = self.csr.col.long()
        self.rows = row_ids(self.csr)
        self.p = None if edge_p is None else edge_p[self.csr.perm.long()]


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
        self._memo = None
        self.reset_parameters()

    def reset_parameters(self):
        self.lin_b.reset_parameters()
        self.lin_u.reset_parameters()
        nn.init.xavier_normal_(self.alpha_b.weight)
        nn.init.xavier_normal_(self.alpha_u.weight)

    def _graphs(self, pos, neg, n):
        key = (pos._version, neg._version, n)
        m = self._memo
        if m is not None and m[0] is pos and m[1] is neg and m[2] == key:
            return m[3]
        if self.first_aggr:
            out = (_Graph(_loop_free_plus_loops(pos), None, n), _Graph(_loop_free_plus_loops(neg), None, n))
        else:
            e1 = _loop_free_plus_loops(pos)
            e2 = neg[:, neg[0] != neg[1]]
            flags = torch.cat([torch.zeros(e1.size(1), dtype=torch.bool, device=pos.device),
                               torch.ones(e2.size(1), dtype=torch.bool, device=pos.device)])
            out = (_Graph(torch.cat([e1, e2], dim=1), flags, n),)
        self._memo = (pos, neg, key, out)
        return out

    def _aggregate(self, g: _Graph, x1, x2, alpha_func):
        w = alpha_func.weight[0]
        a_src, a_dst = w[:self.out_dim], w[self.out_dim:]
        s, d = x1 @ a_src, x1 @ a_dst
        if g.p is None:
            logits = s[g.src] + d[g.rows]
        else:
            s2, d2 = x2 @ a_src, x2 @ a_dst
            logits = torch.where(g.p, s2[g.src], s[g.src]) + torch.where(g.p, d2[g.rows], d[g.rows])
        alpha = segment_softmax(g.csr, torch.tanh(logits + alpha_func.bias))
        if g.p is None:
            return x1 * segment_sum(g.csr, alpha, g.rows).unsqueeze(1)
        share1 = segment_sum(g.csr, alpha * g.p, g.rows)
        share0 = segment_sum(g.csr, alpha * (~g.p), g.rows)
        return x1 * share0.unsqueeze(1) + x2 * share1.unsqueeze(1)

    def forward(self, x: torch.Tensor, pos_edge_index: torch.Tensor, neg_edge_index: torch.Tensor) -> torch.Tensor:
        _cabi.require_gpu(x, pos_edge_index, neg_edge_index)
        graphs = self._graphs(pos_edge_index, neg_edge_index, x.size(0))
        if self.first_aggr:
            h_b, h_u = self.lin_b(x), self.lin_u(x)
            out_b = self._aggregate(graphs[0], h_b, h_b, self.alpha_b)
            out_u = self._aggregate(graphs[1], h_u, h_u, self.alpha_u)
        else:
            h_b, h_u = x[..., :self.in_dim], x[..., self.in_dim:]
            out_b = self._aggregate(graphs[0], self.lin_b(h_b), self.lin_b(h_u), self.alpha_b)
            out_u = self._aggregate(graphs[0], self.lin_u(h_u), self.lin_u(h_b), self.alpha_u)
        return torch.cat([out_b, out_u], dim=-1)

    def __repr__(self) -> str:
        return f'{self.__class__.__name__}({self.in_dim}, {self.out_dim}, first_aggr={self.first_aggr})'
