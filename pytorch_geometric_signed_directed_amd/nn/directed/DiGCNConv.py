"""DiGCNConv -- drop-in for torch_geometric_signed_directed/nn/directed/DiGCNConv.py:9:
out = S^T (x W) + b with the pre-normalised operator S handed in as (edge_index, edge_weight).

Contract kept from the reference (DiGCNConv.py:30-94): constructor arguments, `weight` / `bias` parameters
(glorot / zeros), the `cached_result` / `cached_num_edges` attributes, a cached layer silently keeping the FIRST
operator it saw and raising on a changed edge count, the refusal to run without weights, `__repr__`.
"""
from typing import Optional

import torch
from torch.nn import Parameter

from ... import _cabi
from ...dense import tall_linear
from ...message_passing import MessagePassing
from ...sparse import GLOBAL_PATTERNS, Pattern, spmm
from .._magnetic import glorot, zeros

_STALE_CACHE = ('Cached {} number of edges, but found {}. Please disable the caching behavior of this layer by '
                'removing the `cached=True` argument in its constructor.')
_NO_OPERATOR = 'Normalized adj matrix cannot be None. Please obtain the adj matrix in preprocessing.'


class DiGCNConv(MessagePassing):
    edge_weight_arg = "norm"
    _fused_message = True

    def __init__(self, in_channels: int, out_channels: int, improved: bool = False, cached: bool = True,
                 bias: bool = True, **kwargs):
        super().__init__(aggr='add', **kwargs)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.improved, self.cached = improved, cached
        self.weight = Parameter(torch.Tensor(in_channels, out_channels))
        self.register_parameter('bias', Parameter(torch.Tensor(out_channels)) if bias else None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        zeros(self.bias)
        self.cached_result = self.cached_num_edges = self._pattern = None

    # -- the operator this call aggregates with --------------------------------------------------------
    def _operator(self, edge_index, edge_weight, num_nodes):
        reuse = self.cached and self.cached_result is not None
        if reuse:
            if edge_index.size(1) != self.cached_num_edges:
                raise RuntimeError(_STALE_CACHE.format(self.cached_num_edges, edge_index.size(1)))
            return self._pattern, self.cached_result[1]
        self.cached_num_edges = edge_index.size(1)
        if edge_weight is None:
            raise RuntimeError(_NO_OPERATOR)
        self.cached_result = (edge_index, edge_weight)
        # a cached layer owns its grouping; an uncached one looks the grouping of an unmodified edge_index tensor
        # up (identity + in-place version) instead of re-sorting the edges on every call like the reference
        self._pattern = (Pattern(edge_index, num_nodes, num_nodes, self.flow) if self.cached
                         else GLOBAL_PATTERNS.get(edge_index, num_nodes, num_nodes, self.flow))
        return self._pattern, edge_weight

    def aggregate_projected(self, xw: torch.Tensor, edge_index: torch.Tensor,
                            edge_weight: Optional[torch.Tensor]) -> torch.Tensor:
        """S^T xw + b for features that already went through the weight (callers that batch several layers'
        projections into one GEMM)."""
        pattern, norm = self._operator(edge_index, edge_weight, xw.size(self.node_dim))
        return self.update(spmm(pattern, xw, norm))

    def forward(self, x: torch.FloatTensor, edge_index: torch.LongTensor,
                edge_weight: torch.FloatTensor = None) -> torch.FloatTensor:
        _cabi.require_gpu(x, edge_index, edge_weight)
        if x.dim() == 2:
            projected = tall_linear(x, self.weight)
        else:                                   # [..., N, F] batches: W is shared, so the batch is one tall product
            projected = tall_linear(x.reshape(-1, x.size(-1)), self.weight).view(*x.shape[:-1], self.weight.size(1))
        return self.aggregate_projected(projected, edge_index, edge_weight)

    # -- MessagePassing hooks (generic propagate path) -------------------------------------------------
    def message(self, x_j, norm):
        return x_j if norm is None else norm.view(-1, 1) * x_j

    def update(self, aggr_out):
        return aggr_out if self.bias is None else aggr_out + self.bias

    def __repr__(self):
        return f'{type(self).__name__}({self.in_channels}, {self.out_channels})'
