"""PyG utils used on the path (semantics: SURVEY.md Appendix A)."""
import torch
from .num_nodes import maybe_num_nodes


def _broadcast(index, ref, dim):
    shape = [1] * ref.dim()
    shape[dim] = -1
    return index.view(shape).expand_as(ref)


def scatter(src, index, dim=0, dim_size=None, reduce='sum'):
    dim = src.dim() + dim if dim < 0 else dim
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    size = list(src.size())
    size[dim] = dim_size
    if reduce in ('sum', 'add'):
        return src.new_zeros(size).scatter_add_(dim, _broadcast(index, src, dim), src)
    if reduce == 'mean':
        count = src.new_zeros(dim_size)
        count.scatter_add_(0, index, src.new_ones(src.size(dim)))
        count = count.clamp(min=1)
        out = src.new_zeros(size).scatter_add_(dim, _broadcast(index, src, dim), src)
        return out / _broadcast(count, out, dim)
    if reduce in ('max', 'min'):
        out = src.new_zeros(size)
        return out.scatter_reduce_(dim, _broadcast(index, src, dim), src,
                                   reduce='a' + reduce, include_self=False)
    raise ValueError(reduce)


def remove_self_loops(edge_index, edge_attr=None):
    mask = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, mask]
    return edge_index, (None if edge_attr is None else edge_attr[mask])


def add_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    N = maybe_num_nodes(edge_index, num_nodes)
    loop = torch.arange(N, dtype=edge_index.dtype, device=edge_index.device)
    loop = loop.unsqueeze(0).repeat(2, 1)
    if edge_attr is not None:
        fv = 1. if fill_value is None else fill_value
        loop_attr = edge_attr.new_full((N,) + tuple(edge_attr.shape[1:]), fv)
        edge_attr = torch.cat([edge_attr, loop_attr], dim=0)
    return torch.cat([edge_index, loop], dim=1), edge_attr


def add_remaining_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    N = maybe_num_nodes(edge_index, num_nodes)
    mask = edge_index[0] != edge_index[1]
    loop = torch.arange(N, dtype=edge_index.dtype, device=edge_index.device)
    loop = loop.unsqueeze(0).repeat(2, 1)
    if edge_attr is not None:
        fv = 1. if fill_value is None else fill_value
        loop_attr = edge_attr.new_full((N,) + tuple(edge_attr.shape[1:]), fv)
        inv = ~mask
        loop_attr[edge_index[0][inv]] = edge_attr[inv]
        edge_attr = torch.cat([edge_attr[mask], loop_attr], dim=0)
    return torch.cat([edge_index[:, mask], loop], dim=1), edge_attr


def coalesce(edge_index, edge_attr='???', num_nodes=None, reduce='sum', is_sorted=False,
             sort_by_row=True):
    nnz = edge_index.size(1)
    N = maybe_num_nodes(edge_index, num_nodes)
    key = edge_index[1 - int(sort_by_row)] * N + edge_index[int(sort_by_row)]
    key, perm = key.sort(stable=True)
    edge_index = edge_index[:, perm]
    has_attr = isinstance(edge_attr, torch.Tensor)
    if has_attr:
        edge_attr = edge_attr[perm]
    head = torch.ones(nnz, dtype=torch.bool, device=key.device)
    if nnz > 1:
        head[1:] = key[1:] > key[:-1]
    edge_index = edge_index[:, head]
    if has_attr:
        seg = head.long().cumsum(0) - 1
        edge_attr = scatter(edge_attr, seg, 0, edge_index.size(1), reduce)
        return edge_index, edge_attr
    if edge_attr is None:
        return edge_index, None
    return edge_index


def to_undirected(edge_index, edge_attr='???', num_nodes=None, reduce='add'):
    row, col = edge_index[0], edge_index[1]
    ei = torch.stack([torch.cat([row, col]), torch.cat([col, row])], 0)
    if isinstance(edge_attr, torch.Tensor):
        edge_attr = torch.cat([edge_attr, edge_attr], 0)
    return coalesce(ei, edge_attr, num_nodes, reduce)


def is_undirected(edge_index, edge_attr=None, num_nodes=None):
    N = maybe_num_nodes(edge_index, num_nodes)
    a = set((edge_index[0] * N + edge_index[1]).tolist())
    b = set((edge_index[1] * N + edge_index[0]).tolist())
    return a == b


def to_scipy_sparse_matrix(edge_index, edge_attr=None, num_nodes=None):
    import scipy.sparse as sp
    import numpy as np
    row, col = edge_index.cpu().numpy()
    if edge_attr is None:
        edge_attr = np.ones(row.shape[0])
    else:
        edge_attr = edge_attr.detach().cpu().numpy().reshape(-1)
    N = maybe_num_nodes(edge_index, num_nodes)
    return sp.coo_matrix((edge_attr, (row, col)), (N, N))


def softmax(src, index, ptr=None, num_nodes=None, dim=0):
    N = maybe_num_nodes(index, num_nodes)
    mx = scatter(src.detach(), index, dim, N, 'max').index_select(dim, index)
    out = (src - mx).exp()
    den = scatter(out, index, dim, N, 'sum') + 1e-16
    return out / den.index_select(dim, index)


def negative_sampling(*a, **k):
    raise NotImplementedError("negative_sampling: data-prep, out of scope for the shim")


def structured_negative_sampling(*a, **k):
    raise NotImplementedError


def spmm(*a, **k):
    raise NotImplementedError("SparseTensor path is out of scope (torch_sparse absent)")
