"""DGCNConv -- drop-in for torch_geometric_signed_directed/nn/directed/DGCNConv.py:11 (edge_index
path; the torch_sparse.SparseTensor path is out of scope: torch_sparse is not part of this stack)."""
from typing import Optional

from torch import Tensor

from ... import _cabi, memo
from ...memo import TensorMemo
from ...message_passing import MessagePassing
from ...sparse import Pattern, spmm
from ...utils._norm import gcn_norm


class DGCNConv(MessagePassing):
    edge_weight_arg = "edge_weight"
    _fused_message = True

    def __init__(self, improved: bool = False, cached: bool = False, add_self_loops: bool = True,
                 normalize: bool = True, **kwargs):
        self._memo_switch = kwargs.pop('operator_memo', None)     # memo.py: False = re-normalise every call
        kwargs.setdefault('aggr', 'add')
        super().__init__(**kwargs)
        self.improved = improved
        self.cached = cached
        self.add_self_loops = add_self_loops
        self.normalize = normalize
        self._cached_edge_index = None
        self._cached_adj_t = None
        self.reset_parameters()

    def reset_parameters(self):
        self._cached_edge_index = None
        self._cached_adj_t = None
        self._cached_pattern = None
        self._norm_memo = TensorMemo(6, getattr(self, '_memo_switch', None))

    def forward(self, x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor] = None) -> Tensor:
        if not isinstance(edge_index, Tensor):
            raise NotImplementedError("DGCNConv: only the edge_index (Tensor) path exists on the HIP "
                                      "stack; torch_sparse.SparseTensor is not supported")
        _cabi.require_gpu(x, edge_index, edge_weight)
        n = x.size(self.node_dim)
        pattern = None
        if self.normalize:
            cache = self._cached_edge_index
            if cache is None:
                hit = None if self.cached else self._memo_lookup(edge_index, edge_weight, n)
                if hit is not None:
                    return spmm(hit[1], x, hit[0])
                src_index, src_weight = edge_index, edge_weight
                edge_index, edge_weight = gcn_norm(edge_index, edge_weight, n, self.improved,
                                                   self.add_self_loops, x.dtype)
                # gcn_norm range-checked the ids it was given: its output needs no second device -> host read
                if not self.cached and not (src_weight is not None and src_weight.requires_grad):
                    memo.own(*[t for t in (edge_index, edge_weight) if t is not src_index and t is not src_weight])
                    pattern = Pattern(edge_index, n, n, self.flow, validate=False)
                    self._memo_store(src_index, src_weight, n, edge_weight, pattern)
                if self.cached:
                    # one shared instance called with several operators keeps the FIRST one
                    # (DGCNConv.py:71-81)
                    self._cached_edge_index = (edge_index, edge_weight)
                    self._cached_pattern = Pattern(edge_index, n, n, self.flow, validate=False)
                    pattern = self._cached_pattern
            else:
                edge_index, edge_weight = cache[0], cache[1]
                pattern = self._cached_pattern
        if pattern is None:
            pattern = Pattern(edge_index, n, n, self.flow, validate=not self.normalize)
        return spmm(pattern, x, edge_weight)

    # cached=False (the default) re-normalises and re-sorts on every call in the reference (DGCNConv.py:71-81).
    # gcn_norm is a pure function of (edge_index, edge_weight, N): the results for the last few operators are
    # kept while the inputs are the same tensor objects at the same in-place version (memo.TensorMemo, weakly held;
    # `operator_memo=False` / PYGSD_NO_OPERATOR_MEMO=1 switch it off) -- DGCN_node_classification runs three
    # operators through one instance, twice per forward.
    def _memo_lookup(self, edge_index, edge_weight, n):
        return self._norm_memo.get((edge_index, edge_weight), n)

    def _memo_store(self, edge_index, edge_weight, n, norm_weight, pattern):
        self._norm_memo.put((edge_index, edge_weight), n, (norm_weight, pattern))

    def message(self, x_j: Tensor, edge_weight: Optional[Tensor]) -> Tensor:
        return x_j if edge_weight is None else edge_weight.view(-1, 1) * x_j
