"""Objectives of the signed-embedding callers (SGCN.loss, SDGNN / SiGAT): per-edge dense arithmetic on the
embeddings the hot path produced.  Reference: utils/signed/link_sign_loss.py (Link_Sign_Entropy_Loss :163-230,
Sign_Structure_Loss :233-275, Link_Sign_Product_Loss :130-160, Sign_Product_Entropy_Loss :104-127,
Sign_Direction_Loss :55-100).

The reference draws its "no edge" samples with torch_geometric.utils.negative_sampling /
structured_negative_sampling (random; absent here).  They are restated on the device below with the same
contract -- negative_sampling: as many uniformly drawn (i, j) pairs as edges, none of them a listed edge;
structured_negative_sampling: for every edge (i, j) one k with (i, k) not listed -- and every loss takes the
sampled indices as an optional argument, so the arithmetic is testable independently of the random draw.
"""
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...dense import matmul
from ...memo import TensorMemo

Tensor = torch.Tensor


def _project(z: Tensor, weight: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """z @ weight^T (+ bias) for a tall z [N, d] and a weight with only a few output rows: the libraries' GEMMs for
    1..6 output columns run at ~100 GB/s (1.2 ms at N = 5 * 10^5, d = 64); on the device the product goes through the
    generic HIP GEMM (dense.matmul), which takes any shape."""
    out = matmul(z, weight.t()) if z.is_cuda else z @ weight.t()
    return out if bias is None else out + bias


def _rows_of(x: Tensor, edge_index: Tensor, row: int, cached: bool = True) -> Tensor:
    """x[edge_index[row]]; on the device its backward is a segment reduce over the edge list's CSR
    (sparse.gather_rows) instead of torch's sort-per-call index backward."""
    if x.is_cuda:
        from ...sparse import gather_rows
        return gather_rows(x, edge_index, row, cached)
    return x[edge_index[row]]


_KEY_MEMO = TensorMemo(6)    # edge_index -> sorted unique keys: the samplers are called every training step


def _edge_keys(edge_index: Tensor, n: int) -> Tensor:
    """Sorted unique keys i * n + j of an edge list; kept for the last few edge_index tensors (memo.TensorMemo:
    weakly held identity + in-place version), so a training loop sorts each graph once, not on every loss
    evaluation."""
    keys = _KEY_MEMO.get((edge_index,), n)
    if keys is None:
        keys = _KEY_MEMO.put((edge_index,), n, torch.unique(edge_index[0] * n + edge_index[1]))
    return keys


def _listed(key_sets, query: Tensor) -> Tensor:
    """query keys that occur in any of the sorted key arrays."""
    hit = torch.zeros_like(query, dtype=torch.bool)
    for keys in key_sets:
        if keys.numel():
            pos = torch.searchsorted(keys, query).clamp_(max=keys.numel() - 1)
            hit |= keys[pos] == query
    return hit


def negative_sampling(edge_index, num_nodes: int, num_neg_samples: Optional[int] = None,
                      generator: Optional[torch.Generator] = None) -> Tensor:
    """Uniform (i, j) pairs that are not listed edges; [2, <= num_neg_samples] (fewer only on near-complete
    graphs, as in PyG).  `edge_index` may be one [2, E] tensor or a tuple of them (their union is excluded;
    passing the parts instead of a fresh concatenation lets the sorted keys be reused across calls)."""
    parts = (edge_index,) if isinstance(edge_index, Tensor) else tuple(edge_index)
    want = sum(p.size(1) for p in parts) if num_neg_samples is None else int(num_neg_samples)
    key_sets = [_edge_keys(p, num_nodes) for p in parts]
    dev = parts[0].device
    total = num_nodes * num_nodes
    out = torch.empty(0, dtype=torch.long, device=dev)
    for _ in range(8):
        need = want - out.numel()
        if need <= 0:
            break
        cand = torch.randint(0, total, (int(need * 1.2) + 8,), device=dev, generator=generator)
        cand = cand[~_listed(key_sets, cand)]
        out = torch.unique(torch.cat([out, cand]))          # PyG also returns distinct pairs
    out = out[torch.randperm(out.numel(), device=dev, generator=generator)][:want]
    return torch.stack([out // num_nodes, out % num_nodes])


def structured_negative_sampling(edge_index: Tensor, num_nodes: int, generator: Optional[torch.Generator] = None
                                 ) -> Tuple[Tensor, Tensor, Tensor]:
    """(i, j, k): for every edge (i, j) a node k such that (i, k) is not a listed edge (self pairs allowed,
    PyG's default contains_neg_self_loops=True)."""
    i, j = edge_index[0], edge_index[1]
    key_sets = [_edge_keys(edge_index, num_nodes)]
    k = torch.randint(0, num_nodes, (i.numel(),), device=i.device, generator=generator)
    for _ in range(32):
        bad = _listed(key_sets, i * num_nodes + k)
        idx = bad.nonzero(as_tuple=True)[0]
        if idx.numel() == 0:
            break
        k[idx] = torch.randint(0, num_nodes, (idx.numel(),), device=i.device, generator=generator)
    return i, j, k


class Link_Sign_Entropy_Loss(nn.Module):
    """Three-way (positive / negative / no edge) discriminator loss of SGCN and SNEA."""

    def __init__(self, emb_dim: int) -> None:
        super().__init__()
        self.lin = nn.Linear(2 * emb_dim, 3)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin.reset_parameters()

    def _node_scores(self, z: Tensor):
        """lin([z_i, z_j]) = z_i W_1^T + z_j W_2^T + b: the two halves of the Linear are applied per NODE (N rows)
        and only the 3-wide results are gathered per edge -- the reference gathers both 64-wide rows per edge and
        runs the Linear over E rows (same value, re-associated)."""
        d = z.size(1)
        w = self.lin.weight
        return _project(z, w[:, :d]), _project(z, w[:, d:], self.lin.bias)

    def discriminate(self, z: Tensor, edge_index: Tensor, scores=None, cached: bool = True) -> Tensor:
        src, dst = self._node_scores(z) if scores is None else scores
        return torch.log_softmax(_rows_of(src, edge_index, 0, cached) + _rows_of(dst, edge_index, 1, cached), dim=1)

    def forward(self, z: Tensor, pos_edge_index: Tensor, neg_edge_index: Tensor,
                none_edge_index: Optional[Tensor] = None) -> Tensor:
        if none_edge_index is None:
            none_edge_index = negative_sampling((pos_edge_index, neg_edge_index), z.size(0))
        scores = self._node_scores(z)
        nll = 0
        for label, ei in enumerate((pos_edge_index, neg_edge_index, none_edge_index)):
            # F.nll_loss against a constant label = minus the mean of that column
            nll = nll - self.discriminate(z, ei, scores, cached=label < 2)[:, label].mean()
        return nll / 3.0


class Sign_Structure_Loss(nn.Module):
    """Triplet terms of SGCN: linked-positive pairs closer than sampled non-neighbours, linked-negative pairs
    farther."""

    @staticmethod
    def _triplet(z: Tensor, edge_index: Tensor, k: Optional[Tensor]):
        if k is None:
            _, _, k = structured_negative_sampling(edge_index, z.size(0))
        zi, zj = _rows_of(z, edge_index, 0), _rows_of(z, edge_index, 1)
        zk = _rows_of(z, torch.stack([edge_index[0], k]), 1, cached=False)      # fresh samples: one-off grouping
        return (zi - zj).pow(2).sum(dim=1), (zi - zk).pow(2).sum(dim=1)

    def pos_embedding_loss(self, z: Tensor, pos_edge_index: Tensor, k: Optional[Tensor] = None) -> Tensor:
        d_edge, d_sample = self._triplet(z, pos_edge_index, k)
        return torch.clamp(d_edge - d_sample, min=0).mean()

    def neg_embedding_loss(self, z: Tensor, neg_edge_index: Tensor, k: Optional[Tensor] = None) -> Tensor:
        d_edge, d_sample = self._triplet(z, neg_edge_index, k)
        return torch.clamp(d_sample - d_edge, min=0).mean()

    def forward(self, z: Tensor, pos_edge_index: Tensor, neg_edge_index: Tensor) -> Tensor:
        return self.pos_embedding_loss(z, pos_edge_index) + self.neg_embedding_loss(z, neg_edge_index)


def _edge_dots(z: Tensor, edge_index: Tensor) -> Tensor:
    return (_rows_of(z, edge_index, 0) * _rows_of(z, edge_index, 1)).sum(dim=1)


class Link_Sign_Product_Loss(nn.Module):
    """SiGAT's product loss: -sum logsigmoid(<z_i, z_j>) on positive, -C sum logsigmoid(-<z_i, z_j>) on
    negative edges, C = |E+| / |E-|."""

    def forward(self, z: Tensor, pos_edge_index: Tensor, neg_edge_index: Tensor) -> Tensor:
        loss_pos = -F.logsigmoid(_edge_dots(z, pos_edge_index)).sum()
        loss_neg = -F.logsigmoid(-_edge_dots(z, neg_edge_index)).sum()
        return loss_pos + loss_neg * (pos_edge_index.shape[1] / neg_edge_index.shape[1])


class Sign_Product_Entropy_Loss(nn.Module):
    """SDGNN's sign loss: binary cross entropy of <z_i, z_j> against the edge sign, summed."""

    def forward(self, z: Tensor, pos_edge_index: Tensor, neg_edge_index: Tensor) -> Tensor:
        p1, p2 = _edge_dots(z, pos_edge_index), _edge_dots(z, neg_edge_index)
        return (F.binary_cross_entropy_with_logits(p1, torch.ones_like(p1), reduction='sum')
                + F.binary_cross_entropy_with_logits(p2, torch.zeros_like(p2), reduction='sum'))


class Sign_Direction_Loss(nn.Module):
    """SDGNN's direction loss: hinge-like squared penalty on the status-score difference s1(z_i) - s2(z_j)."""

    def __init__(self, emb_dim: int) -> None:
        super().__init__()
        self.score_function1 = nn.Sequential(nn.Linear(emb_dim, 1), nn.Sigmoid())
        self.score_function2 = nn.Sequential(nn.Linear(emb_dim, 1), nn.Sigmoid())

    def forward(self, z: Tensor, pos_edge_index: Tensor, neg_edge_index: Tensor) -> Tensor:
        s1, s2 = self.score_function1(z), self.score_function2(z)       # per node (N rows), then gathered per edge
        d = _rows_of(s1, pos_edge_index, 0) - _rows_of(s2, pos_edge_index, 1)
        q = torch.where(d > -0.5, torch.full_like(d, -0.5), d)
        pos_loss = (q - d).pow(2).sum()
        d = _rows_of(s1, neg_edge_index, 0) - _rows_of(s2, neg_edge_index, 1)
        q = torch.where(d > 0.5, d, torch.full_like(d, 0.5))
        return pos_loss + (q - d).pow(2).sum()


class Sign_Triangle_Loss(nn.Module):
    """SDGNN's triangle loss (link_sign_loss.py:10-51): BCE of lin([z_i, z_j]) against the edge sign, every edge
    weighted by the number of balanced triangles it closes.  `edge_weight` is the reference's scipy matrix
    (any format); the per-edge weights are looked up ONCE per edge list and kept on the device (the reference
    indexes the scipy matrix with Python lists on every call)."""

    def __init__(self, emb_dim: int, edge_weight) -> None:
        super().__init__()
        self.lin = nn.Linear(emb_dim * 2, 1)
        self.edge_weight = edge_weight
        self._memo = TensorMemo(4)      # per edge list and in-place version (weakly held; memo.py's opt-outs and strict mode)

    def _weights(self, edge_index: Tensor, device) -> Tensor:
        hit = self._memo.get((edge_index,), "triangle weights")
        if hit is not None:
            return hit
        import numpy as np
        ij = edge_index.detach().cpu().numpy()
        w = np.asarray(self.edge_weight.tocsr()[ij[0], ij[1]]).reshape(-1, 1)
        w = torch.from_numpy(w).to(device)
        return self._memo.put((edge_index,), "triangle weights", w)

    def forward(self, z: Tensor, pos_edge_index: Tensor, neg_edge_index: Tensor) -> Tensor:
        dim = z.size(1)                      # lin([z_i, z_j]) = z_i W_1^T + (z_j W_2^T + b), evaluated per node
        head, tail = _project(z, self.lin.weight[:, :dim]), _project(z, self.lin.weight[:, dim:], self.lin.bias)
        rs1 = _rows_of(head, pos_edge_index, 0) + _rows_of(tail, pos_edge_index, 1)
        rs2 = _rows_of(head, neg_edge_index, 0) + _rows_of(tail, neg_edge_index, 1)
        w1, w2 = self._weights(pos_edge_index, z.device), self._weights(neg_edge_index, z.device)
        pos_loss = F.binary_cross_entropy_with_logits(rs1, torch.ones_like(rs1), weight=w1.to(rs1.dtype), reduction='sum')
        neg_loss = F.binary_cross_entropy_with_logits(rs2, torch.zeros_like(rs2), weight=w2.to(rs2.dtype), reduction='sum')
        return pos_loss + neg_loss
