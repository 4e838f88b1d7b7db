"""DiGCL -- directed graph contrastive learning (reference nn/directed/DiGCL.py:8-168).  Its encoder is a stack
of `torch_geometric.nn.GCNConv`; PyG is not part of this stack, so GCNConv (default options) is restated over
the device path: `gcn_norm` (pygsd_self_loops_*, pygsd_csr_row_sum_f32, pygsd_degree_scale_f32), the CSR SpMM,
and the split-K `tall_linear`.  State-dict keys match PyG's (`lin.weight`, `bias`)."""
import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import _cabi
from ...dense import tall_linear
from ...memo import TensorMemo
from ...sparse import Pattern, spmm
from ...utils._norm import gcn_norm


class GCNConv(nn.Module):
    """x' = D^-1/2 (A + I) D^-1/2 (x W) + b (degree over the target column, remaining self loops of weight 1 --
    2 with `improved` and explicit weights).  `cached=False` recomputes the normalisation per call in PyG; here
    the result for unmodified graph tensors is reused (same values)."""

    def __init__(self, in_channels: int, out_channels: int, improved: bool = False, cached: bool = False,
                 add_self_loops: bool = True, normalize: bool = True, bias: bool = True, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.improved, self.cached, self.add_self_loops, self.normalize = improved, cached, add_self_loops, normalize
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self._memo = TensorMemo(4)            # DiGCL alternates two augmented views through one encoder
        self.reset_parameters()

    def reset_parameters(self):
        a = math.sqrt(6.0 / (self.in_channels + self.out_channels))
        self.lin.weight.data.uniform_(-a, a)
        if self.bias is not None:
            self.bias.data.fill_(0)
        self._memo = TensorMemo(4)            # DiGCL alternates two augmented views through one encoder

    def _operator(self, edge_index, edge_weight, n, dtype):
        hit = self._memo.get((edge_index, edge_weight), n)
        if hit is not None:
            return hit
        if self.normalize:
            ei, ew = gcn_norm(edge_index, edge_weight, n, self.improved, self.add_self_loops, dtype)
        else:
            ei, ew = edge_index, edge_weight
        pat = Pattern(ei, n, n)
        if not (edge_weight is not None and edge_weight.requires_grad):
            self._memo.put((edge_index, edge_weight), n, (pat, ew))
        return pat, ew

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor, edge_weight: Optional[torch.Tensor] = None):
        _cabi.require_gpu(x, edge_index, edge_weight)
        pat, ew = self._operator(edge_index, edge_weight, x.size(0), x.dtype)
        out = spmm(pat, tall_linear(x, self.lin.weight.t()), ew)
        return out if self.bias is None else out + self.bias

    def __repr__(self):
        return f'{self.__class__.__name__}({self.in_channels}, {self.out_channels})'


class DiGCL_Encoder(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, activation: str, num_layers: int = 2):
        super().__init__()
        assert num_layers >= 2
        self._num_layers = num_layers
        conv = [GCNConv(in_channels, 2 * out_channels)]
        for _ in range(1, num_layers - 1):
            conv.append(GCNConv(2 * out_channels, 2 * out_channels))
        conv.append(GCNConv(2 * out_channels, out_channels))
        self.conv = nn.ModuleList(conv)
        self.activation = ({'relu': F.relu, 'prelu': nn.PReLU(), 'rrelu': nn.RReLU()})[activation]

    def reset_parameters(self):
        for layer in self.conv:
            layer.reset_parameters()

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor, edge_weight: Optional[torch.Tensor] = None):
        for i in range(self._num_layers):
            x = self.activation(self.conv[i](x, edge_index, edge_weight))
        return x


class DiGCL(nn.Module):
    def __init__(self, in_channels: int, activation: str, num_hidden: int, num_proj_hidden: int, tau: float,
                 num_layers: int):
        super().__init__()
        self.encoder = DiGCL_Encoder(in_channels, num_hidden, activation, num_layers)
        self.tau: float = tau
        self.fc1 = nn.Linear(num_hidden, num_proj_hidden)
        self.fc2 = nn.Linear(num_proj_hidden, num_hidden)

    def reset_parameters(self):
        self.fc1.reset_parameters()
        self.fc2.reset_parameters()
        self.encoder.reset_parameters()

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor, edge_weight: Optional[torch.Tensor] = None):
        return self.encoder(x, edge_index, edge_weight)

    def projection(self, z: torch.Tensor) -> torch.Tensor:
        return self.fc2(F.elu(self.fc1(z)))

    def sim(self, z1: torch.Tensor, z2: torch.Tensor) -> torch.Tensor:
        return torch.mm(F.normalize(z1), F.normalize(z2).t())

    def semi_loss(self, z1: torch.Tensor, z2: torch.Tensor) -> torch.Tensor:
        f = lambda x: torch.exp(x / self.tau)  # noqa: E731
        refl_sim, between_sim = f(self.sim(z1, z1)), f(self.sim(z1, z2))
        return -torch.log(between_sim.diag() / (refl_sim.sum(1) + between_sim.diag() - refl_sim.diag()))

    def batched_semi_loss(self, z1: torch.Tensor, z2: torch.Tensor, batch_size: int) -> torch.Tensor:
        """O(B N) memory instead of O(N^2).  NB: as in the reference (DiGCL.py:125-141) the denominator holds the
        SUM of the cross-view similarities, not only the diagonal, so it differs from `semi_loss`."""
        num_nodes = z1.size(0)
        f = lambda x: torch.exp(x / self.tau)  # noqa: E731
        losses = []
        for i in range((num_nodes - 1) // batch_size + 1):
            rows = slice(i * batch_size, (i + 1) * batch_size)
            refl_sim, between_sim = f(self.sim(z1[rows], z1)), f(self.sim(z1[rows], z2))
            losses.append(-torch.log(between_sim[:, rows].diag()
                                     / (refl_sim.sum(1) + between_sim.sum(1) - refl_sim[:, rows].diag())))
        return torch.cat(losses)

    def loss(self, z1: torch.Tensor, z2: torch.Tensor, mean: bool = True, batch_size: int = 0) -> torch.Tensor:
        h1, h2 = self.projection(z1), self.projection(z2)
        if batch_size == 0:
            l1, l2 = self.semi_loss(h1, h2), self.semi_loss(h2, h1)
        else:
            l1, l2 = self.batched_semi_loss(h1, h2, batch_size), self.batched_semi_loss(h2, h1, batch_size)
        ret = (l1 + l2) * 0.5
        return ret.mean() if mean else ret.sum()
