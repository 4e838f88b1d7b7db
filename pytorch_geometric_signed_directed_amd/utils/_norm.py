"""Device-side operator normalisations used by DGCNConv and Conv_Base:
`add_remaining_self_loops`, `gcn_norm` (torch_geometric.nn.conv.gcn_conv.gcn_norm as called at
reference nn/directed/DGCNConv.py:75) and `conv_norm_rw` (reference nn/general/conv_base.py:12-31).
Element-wise / index arithmetic on GPU tensors; the degree sums are segment reductions."""
from typing import Optional, Tuple

import torch

from .. import _cabi

Tensor = torch.Tensor


def maybe_num_nodes(edge_index: Tensor, num_nodes: Optional[int] = None) -> int:
    if num_nodes is not None:
        return num_nodes
    return int(edge_index.max()) + 1 if edge_index.numel() > 0 else 0


def add_remaining_self_loops(edge_index: Tensor, edge_attr: Optional[Tensor], fill_value: float,
                             num_nodes: int) -> Tuple[Tensor, Optional[Tensor]]:
    """Existing self loops are removed from the list and N loops appended; a node that had a loop
    keeps its weight, the others get `fill_value`."""
    off = edge_index[0] != edge_index[1]
    loops = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device).unsqueeze(0).repeat(2, 1)
    out_index = torch.cat([edge_index[:, off], loops], dim=1)
    if edge_attr is None:
        return out_index, None
    tail = edge_attr.new_full((num_nodes,), fill_value)
    on = ~off
    tail[edge_index[0][on]] = edge_attr[on]
    return out_index, torch.cat([edge_attr[off], tail], dim=0)


def _segment_sum(values: Tensor, index: Tensor, n: int) -> Tensor:
    return torch.zeros(n, dtype=values.dtype, device=values.device).index_add_(0, index, values)


def gcn_norm(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int, improved: bool = False,
             add_self_loops: bool = True, dtype=None) -> Tuple[Tensor, Tensor]:
    _cabi.require_gpu(edge_index, edge_weight)
    if add_self_loops:
        edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight,
                                                           2.0 if improved else 1.0, num_nodes)
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype or torch.float32, device=edge_index.device)
    row, col = edge_index[0], edge_index[1]
    deg = _segment_sum(edge_weight, col, num_nodes)
    dis = deg.pow(-0.5)
    dis = dis.masked_fill(dis == float("inf"), 0)
    return edge_index, dis[row] * edge_weight * dis[col]


def conv_norm_rw(edge_index: Tensor, fill_value: float = 0.5, edge_weight: Optional[Tensor] = None,
                 num_nodes: Optional[int] = None, add_self_loops: bool = True, dtype=None):
    """Random-walk normalisation D^-1 (A + fill I) (conv_base.py:12-31)."""
    _cabi.require_gpu(edge_index, edge_weight)
    num_nodes = maybe_num_nodes(edge_index, num_nodes)
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype or torch.float32, device=edge_index.device)
    if add_self_loops:
        edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, fill_value, num_nodes)
    row = edge_index[0]
    deg = _segment_sum(edge_weight, row, num_nodes)
    inv = deg.pow(-1)
    inv = inv.masked_fill(inv == float("inf"), 0)
    return edge_index, inv[row] * edge_weight
