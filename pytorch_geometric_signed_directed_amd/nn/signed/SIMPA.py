"""SIMPA -- drop-in for torch_geometric_signed_directed/nn/signed/SIMPA.py:11 (signed mixed-path
aggregation of SSSNET): a hop schedule of Conv_Base SpMMs and axpys."""
from typing import Optional

import torch
from torch.nn import Parameter

from ..general.conv_base import Conv_Base, flipped_edge_index


class SIMPA(torch.nn.Module):
    r"""Args mirror the reference (SIMPA.py:20): hop, fill_value, directed=False."""

    def __init__(self, hop: int, fill_value: float, directed: bool = False):
        super().__init__()
        self._hop_p = hop + 1
        self._hop_n = int((1 + hop) * hop / 2)
        self._undirected = not directed
        self.conv_layer_p = Conv_Base(fill_value)
        self.conv_layer_n = Conv_Base(0.0)
        names = ("_w_p", "_w_n") if self._undirected else ("_w_sp", "_w_sn", "_w_tp", "_w_tn")
        for name in names:
            rows = self._hop_n if name.endswith("n") else self._hop_p
            self.register_parameter(name, Parameter(torch.FloatTensor(rows, 1)))
        if self._undirected:
            self._reset_parameters_undirected()
        else:
            self._reset_parameters_directed()

    def _reset_parameters_undirected(self):
        self._w_p.data.fill_(1.0)
        self._w_n.data.fill_(1.0)

    def _reset_parameters_directed(self):
        for name in ("_w_sp", "_w_sn", "_w_tp", "_w_tn"):
            getattr(self, name).data.fill_(1.0)

    def _stream(self, ei_p, w_p, ei_n, w_n, x_pos, x_neg, wp, wn):
        """One (positive, negative) feature pair: feat_p = sum_h wp[h] Ap^h x_pos and the mixed
        paths Ap^m An Ap^h x_neg, in the reference's accumulation order (SIMPA.py:77-93)."""
        feat_p = wp[0] * x_pos
        feat_n = None
        cur_p, aux_n = x_pos, x_neg
        j = 0
        last = self._hop_p - 1
        for h in range(self._hop_p):
            if h > 0:
                cur_p = self.conv_layer_p(cur_p, ei_p, w_p)
                if h != last:      # the reference also advances aux_n at the last hop, but never reads it again
                    aux_n = self.conv_layer_p(aux_n, ei_p, w_p)
                feat_p = torch.addcmul(feat_p, wp[h], cur_p)            # feat_p + wp[h] * cur_p, one pass
            if h != last:
                cur_n = self.conv_layer_n(aux_n, ei_n, w_n)
                feat_n = wn[j] * cur_n if feat_n is None else torch.addcmul(feat_n, wn[j], cur_n)
                j += 1
                for _ in range(self._hop_p - 2 - h):
                    cur_n = self.conv_layer_p(cur_n, ei_p, w_p)
                    feat_n = torch.addcmul(feat_n, wn[j], cur_n)
                    j += 1
        return feat_p, (torch.zeros_like(feat_p) if feat_n is None else feat_n)

    def forward(self, edge_index_p: torch.LongTensor, edge_weight_p: torch.FloatTensor,
                edge_index_n: torch.LongTensor, edge_weight_n: torch.FloatTensor,
                x_p: torch.FloatTensor, x_n: torch.FloatTensor,
                x_pt: Optional[torch.FloatTensor] = None,
                x_nt: Optional[torch.FloatTensor] = None) -> torch.FloatTensor:
        if self._undirected:
            fp, fn = self._stream(edge_index_p, edge_weight_p, edge_index_n, edge_weight_n, x_p, x_n,
                                  self._w_p, self._w_n)
            return torch.cat([fp, fn], dim=1)
        sp, sn = self._stream(edge_index_p, edge_weight_p, edge_index_n, edge_weight_n, x_p, x_n,
                              self._w_sp, self._w_sn)
        tp, tn = self._stream(flipped_edge_index(edge_index_p), edge_weight_p,
                              flipped_edge_index(edge_index_n), edge_weight_n,
                              x_pt, x_nt, self._w_tp, self._w_tn)
        return torch.cat([sp, sn, tp, tn], dim=1)
