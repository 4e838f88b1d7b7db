import torch
from ...utils import add_remaining_self_loops, scatter
from ...utils.num_nodes import maybe_num_nodes


def gcn_norm(edge_index, edge_weight=None, num_nodes=None, improved=False,
             add_self_loops=True, flow='source_to_target', dtype=None):
    """Current-PyG ordering: remaining self loops are added BEFORE the default
    all-ones weights are materialised (so `improved` only matters with weights)."""
    fill_value = 2. if improved else 1.
    num_nodes = maybe_num_nodes(edge_index, num_nodes)
    if add_self_loops:
        edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight,
                                                           fill_value, num_nodes)
    if edge_weight is None:
        edge_weight = torch.ones((edge_index.size(1),), dtype=dtype, device=edge_index.device)
    row, col = edge_index[0], edge_index[1]
    idx = col if flow == 'source_to_target' else row
    deg = scatter(edge_weight, idx, dim=0, dim_size=num_nodes, reduce='sum')
    dis = deg.pow_(-0.5)
    dis.masked_fill_(dis == float('inf'), 0)
    return edge_index, dis[row] * edge_weight * dis[col]
