"""PyG-free `MessagePassing` base with the `propagate(edge_index, size=None, **kwargs)` surface the
reference's conv layers call, routed to the fused HIP SpMM instead of
index_select -> message -> scatter.

Supported message family (everything the reference's layers on this path use):
    message(x_j[, <w>]) = <w>.view(-1, 1) * x_j      (or x_j when <w> is None / absent)
with aggr in {'add', 'sum', 'mean'}, flow in {'source_to_target', 'target_to_source'},
node_dim = -2, `x` a tensor or a (x_src, x_dst) pair, and an `update(aggr_out)` hook.
The per-edge weight keyword is named by the subclass (`edge_weight_arg`, e.g. 'norm').
A subclass that overrides `message` with anything else gets a loud NotImplementedError: there is no
materialise-the-messages fallback.
"""
from typing import Optional

import torch

from .sparse import GLOBAL_PATTERNS, Pattern, spmm


class MessagePassing(torch.nn.Module):
    edge_weight_arg: Optional[str] = "edge_weight"

    def __init__(self, aggr: str = "add", flow: str = "source_to_target", node_dim: int = -2, **kwargs):
        super().__init__()
        if aggr not in ("add", "sum", "mean"):
            raise NotImplementedError(f"aggr={aggr!r} is not on the HIP path (add / mean only)")
        if flow not in ("source_to_target", "target_to_source"):
            raise ValueError(f"unknown flow {flow!r}")
        if node_dim != -2:
            raise NotImplementedError("only node_dim=-2 is supported")
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim

    # hooks kept for API compatibility -------------------------------------------------------
    def message(self, x_j, edge_weight=None):
        return x_j if edge_weight is None else edge_weight.view(-1, 1) * x_j

    def update(self, aggr_out):
        return aggr_out

    # ------------------------------------------------------------------------------------------
    def propagate(self, edge_index, size=None, **kwargs):
        if type(self).message is not MessagePassing.message and not getattr(self, "_fused_message", False):
            raise NotImplementedError(
                f"{type(self).__name__}.message is user-defined; the HIP path only fuses "
                "message = w.view(-1, 1) * x_j (declare `_fused_message = True` if it is that).")
        x = kwargs.get("x")
        if x is None:
            raise ValueError("propagate needs x=...")
        w = kwargs.get(self.edge_weight_arg) if self.edge_weight_arg else None
        s2t = self.flow == "source_to_target"
        if isinstance(x, (tuple, list)):
            x_src, x_dst = x
            x_in = x_src if s2t else x_dst
            n_other = (x_dst if s2t else x_src).size(self.node_dim)
        else:
            x_in, n_other = x, x.size(self.node_dim)
        n_in = x_in.size(self.node_dim)
        n_out = n_other
        if size is not None:
            n_out = size[1] if s2t else size[0]
        if isinstance(edge_index, Pattern):
            pat = edge_index
        else:
            pat = GLOBAL_PATTERNS.get(edge_index, n_in, n_out, self.flow)
        out = spmm(pat, x_in, w, reduce=self.aggr)
        return self.update(out)
