"""DiGCNConv -- drop-in for torch_geometric_signed_directed/nn/directed/DiGCNConv.py:9:
out = S^T (x W) + b with the pre-normalised operator S handed in as (edge_index, edge_weight)."""
import torch
from torch.nn import Parameter

from ... import _cabi
from ...message_passing import MessagePassing
from ...dense import tall_linear
from ...sparse import GLOBAL_PATTERNS, Pattern, spmm
from .._magnetic import glorot, zeros


class DiGCNConv(MessagePassing):
    edge_weight_arg = "norm"
    _fused_message = True

    def __init__(self, in_channels: int, out_channels: int, improved: bool = False, cached: bool = True,
                 bias: bool = True, **kwargs):
        super().__init__(aggr='add', **kwargs)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.improved = improved
        self.cached = cached
        self.weight = Parameter(torch.Tensor(in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        zeros(self.bias)
        self.cached_result = None
        self.cached_num_edges = None
        self._pattern = None

    def forward(self, x: torch.FloatTensor, edge_index: torch.LongTensor,
                edge_weight: torch.FloatTensor = None) -> torch.FloatTensor:
        _cabi.require_gpu(x, edge_index, edge_weight)
        x = tall_linear(x, self.weight) if x.dim() == 2 else torch.matmul(x, self.weight)

        if self.cached and self.cached_result is not None and edge_index.size(1) != self.cached_num_edges:
            raise RuntimeError(
                'Cached {} number of edges, but found {}. Please '
                'disable the caching behavior of this layer by removing '
                'the `cached=True` argument in its constructor.'.format(
                    self.cached_num_edges, edge_index.size(1)))

        if not self.cached or self.cached_result is None:
            self.cached_num_edges = edge_index.size(1)
            if edge_weight is None:
                raise RuntimeError(
                    'Normalized adj matrix cannot be None. Please '
                    'obtain the adj matrix in preprocessing.')
            # cached=True (the default) silently keeps the FIRST operator (DiGCNConv.py:75-85)
            self.cached_result = edge_index, edge_weight
            n = x.size(self.node_dim)
            # cached=False re-groups the edges per call in the reference; the grouping of an unmodified
            # edge_index tensor is looked up instead (identity + in-place version)
            self._pattern = (Pattern(edge_index, n, n, self.flow) if self.cached
                             else GLOBAL_PATTERNS.get(edge_index, n, n, self.flow))

        _, norm = self.cached_result
        return self.update(spmm(self._pattern, x, norm))

    def message(self, x_j, norm):
        return norm.view(-1, 1) * x_j if norm is not None else x_j

    def update(self, aggr_out):
        if self.bias is not None:
            aggr_out = aggr_out + self.bias
        return aggr_out

    def __repr__(self):
        return '{}({}, {})'.format(self.__class__.__name__, self.in_channels, self.out_channels)
