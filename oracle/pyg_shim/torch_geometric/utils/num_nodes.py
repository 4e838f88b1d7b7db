def maybe_num_nodes(edge_index, num_nodes=None):
    if num_nodes is not None:
        return num_nodes
    return int(edge_index.max()) + 1 if edge_index.numel() > 0 else 0
