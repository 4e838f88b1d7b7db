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

Tensor = torch.Tensor


def _edge_keys(edge_index: Tensor, n: int) -> Tensor:
    return torch.unique(edge_index[0] * n + edge_index[1])


def _listed(keys_sorted: Tensor, query: Tensor) -> Tensor:
    if keys_sorted.numel() == 0:
        return torch.zeros_like(query, dtype=torch.bool)
    pos = torch.searchsorted(keys_sorted, query).clamp_(max=keys_sorted.numel() - 1)
    return keys_sorted[pos] == query


def negative_sampling(edge_index: Tensor, num_nodes: int, num_neg_samples: Optional[int] = None,
                      generator: Optional[torch.Generator] = None) -> Tensor:
    """Uniform (i, j) pairs that are not listed edges; [2, <= num_neg_samples] (fewer only on near-complete
    graphs, as in PyG)."""
    want = edge_index.size(1) if num_neg_samples is None else int(num_neg_samples)
    keys = _edge_keys(edge_index, num_nodes)
    dev = edge_index.device
    total = num_nodes * num_nodes
    out = torch.empty(0, dtype=torch.long, device=dev)
    for _ in range(8):
        need = want - out.numel()
        if need <= 0:
            break
        cand = torch.randint(0, total, (int(need * 1.2) + 8,), device=dev, generator=generator)
        cand = cand[~_listed(keys, cand)]
        out = torch.unique(torch.cat([out, cand]))          # PyG also returns distinct pairs
    out = out[torch.randperm(out.numel(), device=dev, generator=generator)][:want]
    return torch.stack([out // num_nodes, out % num_nodes])


def structured_negative_sampling(edge_index: Tensor, num_nodes: int, generator: Optional[torch.Generator] = None
                                 ) -> Tuple[Tensor, Tensor, Tensor]:
    """(i, j, k): for every edge (i, j) a node k such that (i, k) is not a listed edge (self pairs allowed,
    PyG's default contains_neg_self_loops=True)."""
    i, j = edge_index[0], edge_index[1]
    keys = _edge_keys(edge_index, num_nodes)
    k = torch.randint(0, num_nodes, (i.numel(),), device=i.device, generator=generator)
    for _ in range(32):
        bad = _listed(keys, i * num_nodes + k)
        nbad = int(bad.sum())
        if nbad == 0:
            break
        k[bad] = torch.randint(0, num_nodes, (nbad,), device=i.device, generator=generator)
    return i, j, k


class Link_Sign_Entropy_Loss(nn.Module):
    """Three-way (positive / negative / no edge) discriminator loss of SGCN and SNEA."""

    def __init__(self, emb_dim: int) -> None:
        super().__init__()
        self.lin = nn.Linear(2 * emb_dim, 3)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin.reset_parameters()

    def discriminate(self, z: Tensor, edge_index: Tensor) -> Tensor:
        return torch.log_softmax(self.lin(torch.cat([z[edge_index[0]], z[edge_index[1]]], dim=1)), dim=1)

    def forward(self, z: Tensor, pos_edge_index: Tensor, neg_edge_index: Tensor,
                none_edge_index: Optional[Tensor] = None) -> Tensor:
        if none_edge_index is None:
            none_edge_index = negative_sampling(torch.cat([pos_edge_index, neg_edge_index], dim=1), z.size(0))
        nll = 0
        for label, ei in enumerate((pos_edge_index, neg_edge_index, none_edge_index)):
            nll = nll + F.nll_loss(self.discriminate(z, ei), ei.new_full((ei.size(1),), label))
        return nll / 3.0


class Sign_Structure_Loss(nn.Module):
    """Triplet terms of SGCN: linked-positive pairs closer than sampled non-neighbours, linked-negative pairs
    farther."""

    def pos_embedding_loss(self, z: Tensor, pos_edge_index: Tensor, k: Optional[Tensor] = None) -> Tensor:
        i, j = pos_edge_index[0], pos_edge_index[1]
        if k is None:
            i, j, k = structured_negative_sampling(pos_edge_index, z.size(0))
        out = (z[i] - z[j]).pow(2).sum(dim=1) - (z[i] - z[k]).pow(2).sum(dim=1)
        return torch.clamp(out, min=0).mean()

    def neg_embedding_loss(self, z: Tensor, neg_edge_index: Tensor, k: Optional[Tensor] = None) -> Tensor:
        i, j = neg_edge_index[0], neg_edge_index[1]
        if k is None:
            i, j, k = structured_negative_sampling(neg_edge_index, z.size(0))
        out = (z[i] - z[k]).pow(2).sum(dim=1) - (z[i] - z[j]).pow(2).sum(dim=1)
        return torch.clamp(out, min=0).mean()

    def forward(self, z: Tensor, pos_edge_index: Tensor, neg_edge_index: Tensor) -> Tensor:
        return self.pos_embedding_loss(z, pos_edge_index) + self.neg_embedding_loss(z, neg_edge_index)


def _edge_dots(z: Tensor, edge_index: Tensor) -> Tensor:
    return (z[edge_index[0]] * z[edge_index[1]]).sum(dim=1)


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
        d = self.score_function1(z[pos_edge_index[0]]) - self.score_function2(z[pos_edge_index[1]])
        q = torch.where(d > -0.5, torch.full_like(d, -0.5), d)
        pos_loss = (q - d).pow(2).sum()
        d = self.score_function1(z[neg_edge_index[0]]) - self.score_function2(z[neg_edge_index[1]])
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
        self._memo = []

    def _weights(self, edge_index: Tensor, device) -> Tensor:
        for src, ver, w in self._memo:
            if src is edge_index and ver == edge_index._version:
                return w
        import numpy as np
        ij = edge_index.detach().cpu().numpy()
        w = np.asarray(self.edge_weight.tocsr()[ij[0], ij[1]]).reshape(-1, 1)
        w = torch.from_numpy(w).to(device)
        self._memo = (self._memo + [(edge_index, edge_index._version, w)])[-4:]
        return w

    def forward(self, z: Tensor, pos_edge_index: Tensor, neg_edge_index: Tensor) -> Tensor:
        rs1 = self.lin(torch.cat([z[pos_edge_index[0]], z[pos_edge_index[1]]], dim=1))
        rs2 = self.lin(torch.cat([z[neg_edge_index[0]], z[neg_edge_index[1]]], dim=1))
        w1, w2 = self._weights(pos_edge_index, z.device), self._weights(neg_edge_index, z.device)
        pos_loss = F.binary_cross_entropy_with_logits(rs1, torch.ones_like(rs1), weight=w1.to(rs1.dtype), reduction='sum')
        neg_loss = F.binary_cross_entropy_with_logits(rs2, torch.zeros_like(rs2), weight=w2.to(rs2.dtype), reduction='sum')
        return pos_loss + neg_loss
