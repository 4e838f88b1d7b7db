"""DIMPA -- drop-in for torch_geometric_signed_directed/nn/directed/DIMPA.py:9 (directed mixed-path
aggregation of DIGRAC): 2*hop Conv_Base SpMMs."""
import torch
from torch.nn import Parameter

from ..general.conv_base import Conv_Base, flipped_edge_index


class DIMPA(torch.nn.Module):
    def __init__(self, hop: int, fill_value: float = 0.5):
        super().__init__()
        self._hop = hop
        self._w_s = Parameter(torch.FloatTensor(hop + 1, 1))
        self._w_t = Parameter(torch.FloatTensor(hop + 1, 1))
        self.conv_layer = Conv_Base(fill_value)
        self._reset_parameters()

    def _reset_parameters(self):
        self._w_s.data.fill_(1.0)
        self._w_t.data.fill_(1.0)

    def forward(self, x_s: torch.FloatTensor, x_t: torch.FloatTensor, edge_index: torch.FloatTensor,
                edge_weight: torch.FloatTensor) -> torch.FloatTensor:
        feat_s = self._w_s[0] * x_s
        feat_t = self._w_t[0] * x_t
        cur_s, cur_t = x_s, x_t
        edge_index_t = flipped_edge_index(edge_index)
        for h in range(1, 1 + self._hop):
            cur_s = self.conv_layer(cur_s, edge_index, edge_weight)
            cur_t = self.conv_layer(cur_t, edge_index_t, edge_weight)
            feat_s = torch.addcmul(feat_s, self._w_s[h], cur_s)          # feat + w[h] * cur in one pass
            feat_t = torch.addcmul(feat_t, self._w_t[h], cur_t)
        return torch.cat([feat_s, feat_t], dim=1)
