"""DIMPA -- drop-in for torch_geometric_signed_directed/nn/directed/DIMPA.py:9 (directed mixed-path aggregation
of DIGRAC).  Two independent polynomial streams over the random-walk operator A = D^-1 (A + fill I) of
Conv_Base: sources accumulate sum_h w_s[h] A^h x_s, targets the same over the flipped edge list; the result is
their concatenation.  State: `_w_s`, `_w_t` of shape [hop + 1, 1], all ones at reset; one Conv_Base instance
(`conv_layer`) shared by both streams, as in the reference."""
import torch
from torch.nn import Parameter

from ... import memo
from ..general.conv_base import Conv_Base, flipped_edge_index


def _hop_polynomial(step, coefficients: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """coefficients[0] * x + sum_{h >= 1} coefficients[h] * step^h(x); each term is added with one fused
    multiply-add pass (the reference does a multiply and an in-place add per term)."""
    total = coefficients[0] * x
    power = x
    for h in range(1, coefficients.size(0)):
        power = step(power)
        total = torch.addcmul(total, coefficients[h], power)
    return total


class DIMPA(torch.nn.Module):
    def __init__(self, hop: int, fill_value: float = 0.5):
        super().__init__()
        self._hop = hop
        for name in ("_w_s", "_w_t"):
            self.register_parameter(name, Parameter(torch.FloatTensor(hop + 1, 1)))
        self.conv_layer = Conv_Base(fill_value)
        self._reset_parameters()

    def _reset_parameters(self):
        for w in (self._w_s, self._w_t):
            w.data.fill_(1.0)

    def forward(self, x_s: torch.Tensor, x_t: torch.Tensor, edge_index: torch.Tensor,
                edge_weight: torch.Tensor) -> torch.Tensor:
        with memo.verified(edge_index, edge_weight):      # one content check for every memo lookup of this forward
            flipped = flipped_edge_index(edge_index)      # memoised: keeps the operator caches of the flipped list warm
            memo.trust(flipped)                           # (this package's own tensor)
            along = lambda v: self.conv_layer(v, edge_index, edge_weight)       # noqa: E731
            against = lambda v: self.conv_layer(v, flipped, edge_weight)       # noqa: E731
            return torch.cat([_hop_polynomial(along, self._w_s, x_s), _hop_polynomial(against, self._w_t, x_t)], dim=1)
