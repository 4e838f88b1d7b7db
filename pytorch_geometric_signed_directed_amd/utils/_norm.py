"""Device-side operator normalisations used by DGCNConv and Conv_Base:
`add_remaining_self_loops`, `gcn_norm` (torch_geometric.nn.conv.gcn_conv.gcn_norm as called at
reference nn/directed/DGCNConv.py:75) and `conv_norm_rw` (reference nn/general/conv_base.py:12-31),
through csrc/laplacian.hip (pygsd_self_loops_*, pygsd_csr_row_sum_f32, pygsd_degree_scale_f32).
Degrees are sequential row sums over a stable CSR, i.e. summed in the COO order the reference's
scatter_add_ uses; an edge_weight that requires grad takes the differentiable tensor-op route."""
import ctypes
from typing import Optional, Tuple

import torch

from .. import _cabi
from .._cabi import check, ptr, stream_ptr
from ..sparse import csr_from_coo

Tensor = torch.Tensor


def maybe_num_nodes(edge_index: Tensor, num_nodes: Optional[int] = None) -> int:
    if num_nodes is not None:
        return num_nodes
    return int(edge_index.max()) + 1 if edge_index.numel() > 0 else 0


def _f32(w: Optional[Tensor]) -> Optional[Tensor]:
    if w is None:
        return None
    w = w.detach().reshape(-1).contiguous()
    return w if w.dtype == torch.float32 else w.float()


def add_remaining_self_loops(edge_index: Tensor, edge_attr: Optional[Tensor], fill_value: float,
                             num_nodes: int, with_weights: bool = True) -> Tuple[Tensor, Optional[Tensor]]:
    """Existing self loops are removed from the list and N loops appended; a node that had a loop keeps
    the weight of its LAST listed loop, the others get `fill_value`.  With `edge_attr=None` and
    `with_weights=True` the kept edges get weight 1 (the implicit all-ones weights)."""
    _cabi.require_gpu(edge_index, edge_attr)
    dev = edge_index.device
    row, col = edge_index[0].contiguous(), edge_index[1].contiguous()
    e, n = row.numel(), int(num_nodes)
    _cabi.check_node_ids((n, row), (n, col))     # the loop scan indexes last[] by these ids
    w = _f32(edge_attr)
    lib = _cabi.lib()
    with _cabi.on_device(dev):
        need = ctypes.c_size_t(0)
        check(lib.pygsd_self_loops_workspace(e, ctypes.byref(need)), "pygsd_self_loops_workspace")
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        last = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        check(lib.pygsd_self_loops_scan(ptr(row), ptr(col), e, n, ptr(ws), need.value, ptr(last), ptr(count),
                                        stream_ptr()), "pygsd_self_loops_scan")
        kept = int(count.item())
        out = torch.empty((2, kept + n), dtype=torch.int64, device=dev)
        out_w = torch.empty(kept + n, dtype=torch.float32, device=dev) if with_weights else None
        if kept + n:
            check(lib.pygsd_self_loops_emit(ptr(row), ptr(col), ptr(w), e, n, float(fill_value), kept, ptr(ws),
                                            need.value, ptr(last), ptr(out[0]), ptr(out[1]), ptr(out_w),
                                            stream_ptr()), "pygsd_self_loops_emit")
    return out, out_w


def _degree(index_row: Tensor, other: Tensor, w: Tensor, n: int) -> Tensor:
    """deg[r] = sum of w over the entries with index_row == r, in COO order."""
    csr = csr_from_coo(index_row, other, n, n)
    deg = torch.empty(n, dtype=torch.float32, device=w.device)
    with _cabi.on_device(w.device):
        check(_cabi.lib().pygsd_csr_row_sum_f32(ptr(csr.rowptr), ptr(csr.perm), ptr(w), n, ptr(deg), stream_ptr()),
              "pygsd_csr_row_sum_f32")
    return deg


def _scale(edge_index: Tensor, w: Tensor, deg: Tensor, mode: int) -> Tensor:
    out = torch.empty_like(w)
    with _cabi.on_device(w.device):
        check(_cabi.lib().pygsd_degree_scale_f32(ptr(edge_index[0]), ptr(edge_index[1]), ptr(w), ptr(deg),
                                                 w.numel(), mode, ptr(out), stream_ptr()), "pygsd_degree_scale_f32")
    return out


def _differentiable(edge_weight: Optional[Tensor]) -> bool:
    return edge_weight is not None and edge_weight.requires_grad


def _loops_torch(edge_index, edge_attr, fill_value, num_nodes):
    off = edge_index[0] != edge_index[1]
    loops = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device).unsqueeze(0).repeat(2, 1)
    tail = edge_attr.new_full((num_nodes,), fill_value)
    on = ~off
    # a node with several listed loops keeps the LAST one (as the HIP route and the reference do): pick the
    # largest COO position per node, then gather through it -- index_put on repeated indices is unordered
    pos = on.nonzero(as_tuple=True)[0]
    if pos.numel():
        node = edge_index[0][pos]
        last = torch.full((num_nodes,), -1, dtype=torch.long, device=edge_index.device)
        last = last.scatter_reduce(0, node, pos, reduce="amax", include_self=True)
        has = last >= 0
        tail = torch.where(has, edge_attr[last.clamp(min=0)], tail)
    return torch.cat([edge_index[:, off], loops], dim=1), torch.cat([edge_attr[off], tail], dim=0)


def gcn_norm(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int, improved: bool = False,
             add_self_loops: bool = True, dtype=None) -> Tuple[Tensor, Tensor]:
    """D^-1/2 (A + fill I) D^-1/2 with the degree taken over the TARGET column; current-PyG ordering
    (loops are added before default weights are materialised, so `improved` needs explicit weights)."""
    _cabi.require_gpu(edge_index, edge_weight)
    n = int(num_nodes)
    fill = 2.0 if improved else 1.0
    if _differentiable(edge_weight):
        # the tensor-op route below never reaches a checked kernel: range-check here, the callers skip theirs
        _cabi.check_node_ids((n, edge_index[0]), (n, edge_index[1]))
        if add_self_loops:
            edge_index, edge_weight = _loops_torch(edge_index, edge_weight, fill, n)
        row, col = edge_index[0], edge_index[1]
        deg = torch.zeros(n, dtype=edge_weight.dtype, device=edge_weight.device).index_add_(0, col, edge_weight)
        dis = deg.pow(-0.5)
        dis = dis.masked_fill(dis == float("inf"), 0)
        return edge_index, dis[row] * edge_weight * dis[col]
    if add_self_loops:
        edge_index, w = add_remaining_self_loops(edge_index, edge_weight, fill if edge_weight is not None else 1.0, n)
    else:
        edge_index = edge_index.contiguous()
        w = _f32(edge_weight)
        if w is None:
            w = torch.ones(edge_index.size(1), dtype=torch.float32, device=edge_index.device)
    deg = _degree(edge_index[1], edge_index[0], w, n)
    return edge_index, _scale(edge_index, w, deg, 0)


def conv_norm_rw(edge_index: Tensor, fill_value: float = 0.5, edge_weight: Optional[Tensor] = None,
                 num_nodes: Optional[int] = None, add_self_loops: bool = True, dtype=None):
    """Random-walk normalisation D^-1 (A + fill I) (conv_base.py:12-31)."""
    _cabi.require_gpu(edge_index, edge_weight)
    n = maybe_num_nodes(edge_index, num_nodes)
    if _differentiable(edge_weight):
        # the tensor-op route below never reaches a checked kernel: range-check here, the callers skip theirs
        _cabi.check_node_ids((n, edge_index[0]), (n, edge_index[1]))
        if add_self_loops:
            edge_index, edge_weight = _loops_torch(edge_index, edge_weight, fill_value, n)
        row = edge_index[0]
        deg = torch.zeros(n, dtype=edge_weight.dtype, device=edge_weight.device).index_add_(0, row, edge_weight)
        inv = deg.pow(-1)
        inv = inv.masked_fill(inv == float("inf"), 0)
        return edge_index, inv[row] * edge_weight
    if add_self_loops:
        edge_index, w = add_remaining_self_loops(edge_index, edge_weight, fill_value, n)
    else:
        edge_index = edge_index.contiguous()
        w = _f32(edge_weight)
        if w is None:
            w = torch.ones(edge_index.size(1), dtype=torch.float32, device=edge_index.device)
    deg = _degree(edge_index[0], edge_index[1], w, n)
    return edge_index, _scale(edge_index, w, deg, 1)
