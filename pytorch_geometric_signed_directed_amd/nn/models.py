"""Thin restatements of the reference's model classes that are CALLERS of the hot path
(SURVEY.md 2 #11): stacks of the drop-in conv layers plus dense torch ops.  Same constructor
signatures, parameter names and forward returns as the reference, so its checkpoints load and its
example scripts run unchanged; all message passing goes through the HIP layers.

Reference files: nn/directed/MagNet_node_classification.py, MagNet_link_prediction.py,
DiGCN_node_classification.py, DiGCN_Inception_Block.py, DiGCN_Inception_Block_node_classification.py,
DIGRAC_node_clustering.py, nn/general/MSGNN.py, nn/signed/SSSNET_node_clustering.py, the *_link_prediction.py
variants of DGCN / DiGCN / DiGCN_Inception_Block / SSSNET, nn/signed/SGCN.py, SNEA.py, SDGNN.py, SiGAT.py.
"""
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Parameter

from .. import _cabi
from ..dense import column_sums_of, tall_gram, tall_linear, tall_product
from ..sparse import _spmm_raw, spmm_rows_into
from .directed.complex_relu import complex_relu_layer
from .directed.DGCNConv import DGCNConv
from .directed.DiGCNConv import DiGCNConv
from .directed.DIMPA import DIMPA
from .directed.MagNetConv import MagNetConv
from .general.MSConv import MSConv
from .signed.SGCNConv import SGCNConv
from .signed.SIMPA import SIMPA


def _conv1x1(conv: nn.Conv1d, x: torch.Tensor) -> torch.Tensor:
    """The reference's `Conv(x.t().unsqueeze(0))[0].t()` with kernel_size 1 is x W^T + b over the node rows:
    the same parameters as a row-major GEMM (split-K weight gradient) instead of a transposed copy of the
    [N, C] activations and a length-N convolution."""
    return tall_linear(x, conv.weight[:, :, 0].t(), conv.bias)


class _MagneticStack(nn.Module):
    """Chebs (ModuleList of magnetic convs) [+ complex ReLU] shared by the MagNet / MSGNN heads."""

    def _build_stack(self, first, rest, layer, activation, normalization, dropout):
        self.normalization = normalization
        self.activation = activation
        if activation:
            self.complex_relu = complex_relu_layer()
        self.Chebs = nn.ModuleList([first] + [rest() for _ in range(1, layer)])
        self.dropout = dropout

    def _encode(self, real, imag, edge_index, edge_weight):
        for cheb in self.Chebs:
            real, imag = cheb(real, imag, edge_index, edge_weight)
            if self.activation:
                real, imag = self.complex_relu(real, imag)
        return real, imag

    def _node_head(self, real, imag):
        """cat -> dropout -> 1x1 Conv1d over the feature axis -> (z, log-probs [N, C])."""
        x = torch.cat((real, imag), dim=-1)
        if self.dropout > 0:
            x = F.dropout(x, self.dropout, training=self.training)
        return x, F.log_softmax(_conv1x1(self.Conv, x), dim=1)

    def _link_head(self, real, imag, query_edges):
        a, b = query_edges[:, 0], query_edges[:, 1]
        x = torch.cat((real[a], real[b], imag[a], imag[b]), dim=-1)
        if self.dropout > 0:
            x = F.dropout(x, self.dropout, training=self.training)
        return x, F.log_softmax(self.linear(x), dim=1)


class MagNet_node_classification(_MagneticStack):
    """nn/directed/MagNet_node_classification.py:12-92."""

    def __init__(self, num_features: int, hidden: int = 2, q: float = 0.25, K: int = 1, label_dim: int = 2,
                 activation: bool = False, trainable_q: bool = False, layer: int = 2, dropout: float = False,
                 normalization: str = 'sym', cached: bool = False):
        super().__init__()
        kw = dict(K=K, q=q, trainable_q=trainable_q, normalization=normalization, cached=cached)
        self._build_stack(MagNetConv(in_channels=num_features, out_channels=hidden, **kw),
                          lambda: MagNetConv(in_channels=hidden, out_channels=hidden, **kw),
                          layer, activation, normalization, dropout)
        self.Conv = nn.Conv1d(2 * hidden, label_dim, kernel_size=1)

    def reset_parameters(self):
        for cheb in self.Chebs:
            cheb.reset_parameters()
        self.Conv.reset_parameters()

    def forward(self, real, imag, edge_index, edge_weight: Optional[torch.Tensor] = None):
        real, imag = self._encode(real, imag, edge_index, edge_weight)
        return self._node_head(real, imag)[1]


class MagNet_link_prediction(_MagneticStack):
    """nn/directed/MagNet_link_prediction.py:12-89."""

    def __init__(self, num_features: int, hidden: int = 2, q: float = 0.25, K: int = 1, label_dim: int = 2,
                 activation: bool = True, trainable_q: bool = False, layer: int = 2, dropout: float = 0.5,
                 normalization: str = 'sym', cached: bool = False):
        super().__init__()
        kw = dict(K=K, q=q, trainable_q=trainable_q, normalization=normalization, cached=cached)
        self._build_stack(MagNetConv(in_channels=num_features, out_channels=hidden, **kw),
                          lambda: MagNetConv(in_channels=hidden, out_channels=hidden, **kw),
                          layer, activation, normalization, dropout)
        self.linear = nn.Linear(hidden * 4, label_dim)

    def reset_parameters(self):
        for cheb in self.Chebs:
            cheb.reset_parameters()
        self.linear.reset_parameters()

    def forward(self, real, imag, edge_index, query_edges, edge_weight: Optional[torch.Tensor] = None):
        real, imag = self._encode(real, imag, edge_index, edge_weight)
        return self._link_head(real, imag, query_edges)[1]


def _msgnn_convs(num_features, hidden, K, q, trainable_q, normalization, cached, conv_bias, absolute_degree):
    # the reference builds the FIRST MSConv without `cached` / `absolute_degree`
    # (general/MSGNN.py:45-46, :130-131)
    first = MSConv(in_channels=num_features, out_channels=hidden, K=K, q=q, trainable_q=trainable_q,
                   normalization=normalization, bias=conv_bias)
    rest = lambda: MSConv(in_channels=hidden, out_channels=hidden, K=K, q=q, trainable_q=trainable_q,  # noqa: E731
                          normalization=normalization, bias=conv_bias, cached=cached,
                          absolute_degree=absolute_degree)
    return first, rest


class MSGNN_link_prediction(_MagneticStack):
    """nn/general/MSGNN.py:11-91."""

    def __init__(self, num_features: int, hidden: int = 2, q: float = 0.25, K: int = 2, label_dim: int = 2,
                 activation: bool = True, trainable_q: bool = False, layer: int = 2, dropout: float = 0.5,
                 normalization: str = 'sym', cached: bool = False, conv_bias: bool = True,
                 absolute_degree: bool = True):
        super().__init__()
        first, rest = _msgnn_convs(num_features, hidden, K, q, trainable_q, normalization, cached, conv_bias,
                                   absolute_degree)
        self._build_stack(first, rest, layer, activation, normalization, dropout)
        self.linear = nn.Linear(hidden * 4, label_dim)

    def reset_parameters(self):
        for cheb in self.Chebs:
            cheb.reset_parameters()
        self.linear.reset_parameters()

    def forward(self, real, imag, edge_index, query_edges, edge_weight: Optional[torch.Tensor] = None):
        real, imag = self._encode(real, imag, edge_index, edge_weight)
        x, out = self._link_head(real, imag, query_edges)
        self.z = x.clone()
        return out


class MSGNN_node_classification(_MagneticStack):
    """nn/general/MSGNN.py:94-188: returns (normalised z, log-probs, argmax, probs)."""

    def __init__(self, num_features: int, hidden: int = 2, q: float = 0.25, K: int = 2, label_dim: int = 2,
                 activation: bool = False, trainable_q: bool = False, layer: int = 2, dropout: float = False,
                 normalization: str = 'sym', cached: bool = False, conv_bias: bool = True,
                 absolute_degree: bool = True):
        super().__init__()
        first, rest = _msgnn_convs(num_features, hidden, K, q, trainable_q, normalization, cached, conv_bias,
                                   absolute_degree)
        self._build_stack(first, rest, layer, activation, normalization, dropout)
        self.Conv = nn.Conv1d(2 * hidden, label_dim, kernel_size=1)

    def reset_parameters(self):
        for cheb in self.Chebs:
            cheb.reset_parameters()
        self.Conv.reset_parameters()

    def forward(self, real, imag, edge_index, edge_weight: Optional[torch.Tensor] = None):
        real, imag = self._encode(real, imag, edge_index, edge_weight)
        z, output = self._node_head(real, imag)
        return F.normalize(z.clone()), output, torch.argmax(output, dim=1), F.softmax(output, dim=1)


class DiGCN_node_classification(nn.Module):
    """nn/directed/DiGCN_node_classification.py:9-46."""

    def __init__(self, num_features: int, hidden: int, label_dim: int, dropout: float = 0.5):
        super().__init__()
        self.conv1 = DiGCNConv(num_features, hidden)
        self.conv2 = DiGCNConv(hidden, label_dim)
        self.dropout = dropout
        self.reset_parameters()

    def reset_parameters(self):
        self.conv1.reset_parameters()
        self.conv2.reset_parameters()

    def forward(self, x, edge_index, edge_weight=None):
        x = F.relu(self.conv1(x, edge_index, edge_weight))
        x = F.dropout(x, p=self.dropout, training=self.training)
        return F.log_softmax(self.conv2(x, edge_index, edge_weight), dim=1)


class _InceptionBlockFn(torch.autograd.Function):
    """The whole DiGCN inception block as one autograd node (fixed operator values):
        forward   [x0 | P_1 | P_2] = x [W_ln^T | W_1 | W_2] + [b_ln | 0 | 0]    ONE product (csrc/tall.hip: x read once, the
                  Linear's bias in its epilogue, the three blocks written as three contiguous matrices)
                  x_k = S_k^T P_k + b_k                                 the HIP SpMM, the conv bias
                                                                        added in its epilogue (Z row with stride 0)
        backward  dP_k = S_k dx_k  written by the SpMM straight into the column halves of ONE [N, 2F] buffer;
                  dx = [dx0 | dP_1 | dP_2] [W_ln^T | W_1 | W_2]^T   ONE product over the three column segments;
                  [dW_ln | dW_1 | dW_2] = x^T [dx0 | dP_1 | dP_2]   ONE pass (csrc/gram.hip); the bias gradients are column sums.
    Replaces three GEMMs + three bias passes forward and three GEMMs + two gradient-accumulation passes + three skinny
    weight GEMMs backward (reference: DiGCN_Inception_Block.py:44-46, DiGCNConv.py:66,86-93)."""

    @staticmethod
    def forward(ctx, x, w_ln_t, b_ln, w1, b1, w2, b2, pat1, ew1, pat2, ew2):
        f = w1.size(1)
        wcat = torch.cat([w_ln_t, w1, w2], dim=1)
        bcat = None if b_ln is None else torch.cat([b_ln, b_ln.new_zeros(2 * f)])
        # three contiguous matrices: x0 as the caller gets it, and the two the aggregations gather whole rows from
        x0, p1, p2 = tall_product([x], wcat, False, bcat, splits=(f, f, f))
        v1, v2 = pat1.values_for(ew1, "fwd"), pat2.values_for(ew2, "fwd")
        x1 = _spmm_raw(pat1.fwd, v1, p1, None, 1.0, 0.0, False, b1)
        x2 = _spmm_raw(pat2.fwd, v2, p2, None, 1.0, 0.0, False, b2)
        ctx.save_for_backward(x, wcat)
        ctx.ops = (pat1, ew1, pat2, ew2)
        ctx.has_bias = (b_ln is not None, b1 is not None, b2 is not None)
        return x0, x1, x2

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g0, g1, g2):
        x, wcat = ctx.saved_tensors
        pat1, ew1, pat2, ew2 = ctx.ops
        f = wcat.size(1) // 3
        g0, g1, g2 = g0.contiguous(), g1.contiguous(), g2.contiguous()
        dp = torch.empty((x.size(0), 2 * f), dtype=x.dtype, device=x.device)
        spmm_rows_into(pat1.bwd, pat1.values_for(ew1, "bwd"), g1, dp[:, :f])
        spmm_rows_into(pat2.bwd, pat2.values_for(ew2, "bwd"), g2, dp[:, f:])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = tall_product([g0, dp], wcat, True)         # g0 W_ln + [dP_1 | dP_2] [W_1 | W_2]^T in one pass
        dw = tall_gram([x], [g0, dp])                       # x^T [dx0 | dP_1 | dP_2]: x, dx0 and dP read once
        dw_ln_t, dw12 = dw[:, :f], dw[:, f:]
        hb = ctx.has_bias
        db0, db1, db2 = column_sums_of([g0 if hb[0] else None, g1 if hb[1] else None, g2 if hb[2] else None])
        return (dx, dw_ln_t, db0, dw12[:, :f], db1, dw12[:, f:], db2, None, None, None, None)


class DiGCN_InceptionBlock(nn.Module):
    """nn/directed/DiGCN_Inception_Block.py:9-47: x0 = Linear(x), x1 / x2 = DiGCNConv on the first- /
    second-order proximity operators."""

    def __init__(self, in_dim: int, out_dim: int):
        super().__init__()
        self.ln = nn.Linear(in_dim, out_dim)
        self.conv1 = DiGCNConv(in_dim, out_dim)
        self.conv2 = DiGCNConv(in_dim, out_dim)
        self.reset_parameters()

    def reset_parameters(self):
        self.ln.reset_parameters()
        self.conv1.reset_parameters()
        self.conv2.reset_parameters()

    def forward(self, x, edge_index, edge_weight, edge_index2, edge_weight2):
        f = self.conv1.out_channels
        quantum = 8 if x.dtype == torch.bfloat16 else 4        # 16-byte column slices for the vector SpMM
        fused = (x.dim() == 2 and x.is_cuda and f % quantum == 0 and edge_weight is not None and edge_weight2 is not None
                 and not edge_weight.requires_grad and not edge_weight2.requires_grad
                 and x.dtype in (torch.float32, torch.bfloat16) and x.dtype == self.conv1.weight.dtype)
        if not fused:
            x0 = tall_linear(x, self.ln.weight.t(), self.ln.bias) if x.dim() == 2 else self.ln(x)
            return x0, self.conv1(x, edge_index, edge_weight), self.conv2(x, edge_index2, edge_weight2)
        _cabi.require_gpu(x, edge_index, edge_weight, edge_index2, edge_weight2)
        n = x.size(0)
        pat1, ew1 = self.conv1._operator(edge_index, edge_weight, n)     # (cached-operator semantics of DiGCNConv)
        pat2, ew2 = self.conv2._operator(edge_index2, edge_weight2, n)
        return _InceptionBlockFn.apply(x, self.ln.weight.t(), self.ln.bias, self.conv1.weight, self.conv1.bias,
                                       self.conv2.weight, self.conv2.bias, pat1, ew1, pat2, ew2)


class DiGCN_Inception_Block_node_classification(nn.Module):
    """nn/directed/DiGCN_Inception_Block_node_classification.py:9-73: three inception blocks, the three
    branches summed with dropouts."""

    def __init__(self, num_features: int, hidden: int, label_dim: int, dropout: float = 0.5):
        super().__init__()
        self.ib1 = DiGCN_InceptionBlock(num_features, hidden)
        self.ib2 = DiGCN_InceptionBlock(hidden, hidden)
        self.ib3 = DiGCN_InceptionBlock(hidden, label_dim)
        self._dropout = dropout
        self.reset_parameters()

    def reset_parameters(self):
        for ib in (self.ib1, self.ib2, self.ib3):
            ib.reset_parameters()

    def forward(self, features, edge_index_tuple, edge_weight_tuple):
        (ei1, ei2), (ew1, ew2) = edge_index_tuple, edge_weight_tuple
        drop = lambda t: F.dropout(t, p=self._dropout, training=self.training)  # noqa: E731
        x = features
        for depth, ib in enumerate((self.ib1, self.ib2, self.ib3)):
            x0, x1, x2 = ib(x, ei1, ew1, ei2, ew2)
            x = drop(x0) + drop(x1) + drop(x2)
            if depth < 2:
                x = drop(x)
        return F.log_softmax(x, dim=1)


class DGCN_node_classification(nn.Module):
    """nn/directed/DGCN_node_classification.py:10-97: ONE shared weight-less DGCNConv applied to the
    symmetrised, in- and out-proximity operators (so `cached=True` silently reuses the first operator for all
    three), two Linear stages, a 1x1 Conv1d head."""

    def __init__(self, num_features: int, hidden: int, label_dim: int, dropout: Optional[float] = 0.5,
                 improved: bool = False, cached: bool = False):
        super().__init__()
        self.dropout = dropout
        self.dgconv = DGCNConv(improved=improved, cached=cached)
        self.Conv = nn.Conv1d(hidden * 3, label_dim, kernel_size=1)
        self.lin1 = nn.Linear(num_features, hidden, bias=False)
        self.lin2 = nn.Linear(hidden * 3, hidden, bias=False)
        self.bias1 = Parameter(torch.Tensor(1, hidden))
        self.bias2 = Parameter(torch.Tensor(1, hidden))
        nn.init.zeros_(self.bias1)
        nn.init.zeros_(self.bias2)

    def reset_parameters(self):
        self.lin1.reset_parameters()
        self.lin2.reset_parameters()
        nn.init.zeros_(self.bias1)
        nn.init.zeros_(self.bias2)
        self.Conv.reset_parameters()

    def _three(self, x, edge_index, edge_in, edge_out, in_w, out_w, bias):
        parts = [self.dgconv(x, edge_index), self.dgconv(x, edge_in, in_w), self.dgconv(x, edge_out, out_w)]
        return F.relu(torch.cat([t + bias for t in parts], dim=-1))

    def forward(self, x, edge_index, edge_in, edge_out, in_w=None, out_w=None):
        x = tall_linear(x, self.lin1.weight.t())
        x = self._three(x, edge_index, edge_in, edge_out, in_w, out_w, self.bias1)
        x = tall_linear(x, self.lin2.weight.t())
        x = self._three(x, edge_index, edge_in, edge_out, in_w, out_w, self.bias2)
        if self.dropout > 0:
            x = F.dropout(x, self.dropout, training=self.training)
        return F.log_softmax(_conv1x1(self.Conv, x), dim=1)


def _cluster_head(z, w_prob, bias):
    output = tall_linear(z, w_prob)
    if bias is not None:
        output = output + bias
    return F.normalize(z), F.log_softmax(output, dim=1), torch.argmax(output, dim=1), F.softmax(output, dim=1)


class DIGRAC_node_clustering(nn.Module):
    """nn/directed/DIGRAC_node_clustering.py:11-89."""

    def __init__(self, num_features: int, hidden: int, nclass: int, fill_value: float, dropout: float, hop: int):
        super().__init__()
        self._num_clusters = int(nclass)
        self._w_s0 = Parameter(torch.FloatTensor(num_features, hidden))
        self._w_s1 = Parameter(torch.FloatTensor(hidden, hidden))
        self._w_t0 = Parameter(torch.FloatTensor(num_features, hidden))
        self._w_t1 = Parameter(torch.FloatTensor(hidden, hidden))
        self._dimpa = DIMPA(hop, fill_value)
        self._relu = nn.ReLU()
        self.dropout = nn.Dropout(p=dropout)
        self._bias = Parameter(torch.FloatTensor(self._num_clusters))
        self._W_prob = Parameter(torch.FloatTensor(2 * hidden, self._num_clusters))
        self._reset_parameters()

    def _reset_parameters(self):
        for w in (self._w_s0, self._w_s1, self._w_t0, self._w_t1, self._W_prob):
            nn.init.xavier_uniform_(w, gain=1.414)
        self._bias.data.fill_(0.0)

    def _mlp(self, x, w0, w1):
        return tall_linear(self.dropout(self._relu(tall_linear(x, w0))), w1)

    def forward(self, edge_index, edge_weight, features):
        z = self._dimpa(self._mlp(features, self._w_s0, self._w_s1), self._mlp(features, self._w_t0, self._w_t1),
                        edge_index, edge_weight)
        return _cluster_head(z, self._W_prob, self._bias)


class SSSNET_node_clustering(nn.Module):
    """nn/signed/SSSNET_node_clustering.py:11-160."""

    def __init__(self, nfeat: int, hidden: int, nclass: int, dropout: float, hop: int, fill_value: float,
                 directed: bool = False, bias: bool = True):
        super().__init__()
        self._num_clusters = int(nclass)
        self._simpa = SIMPA(hop, fill_value, directed)
        if bias:
            self._bias = Parameter(torch.FloatTensor(self._num_clusters))
        else:
            self.register_parameter('_bias', None)
        self._relu = nn.ReLU()
        self._dropout = nn.Dropout(p=dropout)
        self._undirected = not directed
        self._streams = ("p", "n") if self._undirected else ("sp", "sn", "tp", "tn")
        for s in self._streams:
            setattr(self, f"_w_{s}0", Parameter(torch.FloatTensor(nfeat, hidden)))
            setattr(self, f"_w_{s}1", Parameter(torch.FloatTensor(hidden, hidden)))
        self._W_prob = Parameter(torch.FloatTensor(len(self._streams) * hidden, self._num_clusters))
        self._reset_parameters()

    def _reset_parameters(self):
        for s in self._streams:
            nn.init.xavier_uniform_(getattr(self, f"_w_{s}0"), gain=1.414)
            nn.init.xavier_uniform_(getattr(self, f"_w_{s}1"), gain=1.414)
        if self._bias is not None:
            self._bias.data.fill_(0.0)
        nn.init.xavier_uniform_(self._W_prob, gain=1.414)

    _reset_parameters_undirected = _reset_parameters
    _reset_parameters_directed = _reset_parameters

    def forward(self, edge_index_p, edge_weight_p, edge_index_n, edge_weight_n, features
                ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        xs = [tall_linear(self._dropout(self._relu(tall_linear(features, getattr(self, f"_w_{s}0")))),
                       getattr(self, f"_w_{s}1")) for s in self._streams]
        z = self._simpa(edge_index_p, edge_weight_p, edge_index_n, edge_weight_n, *xs)
        return _cluster_head(z, self._W_prob, self._bias)


# --------------------------------------------------------------------------------------------------
# link-prediction callers: the same encoders, read out on pairs of node rows
# --------------------------------------------------------------------------------------------------
def _pair_rows(x, query_edges):
    return torch.cat((x[query_edges[:, 0]], x[query_edges[:, 1]]), dim=-1)


class DiGCN_link_prediction(nn.Module):
    """nn/directed/DiGCN_link_prediction.py:9-52."""

    def __init__(self, num_features: int, hidden: int, label_dim: int, dropout: float = 0.5):
        super().__init__()
        self.conv1 = DiGCNConv(num_features, hidden)
        self.conv2 = DiGCNConv(hidden, hidden)
        self.dropout = dropout
        self.linear = nn.Linear(hidden * 2, label_dim)
        self.reset_parameters()

    def reset_parameters(self):
        self.conv1.reset_parameters()
        self.conv2.reset_parameters()
        self.linear.reset_parameters()

    def forward(self, x, edge_index, query_edges, edge_weight=None):
        x = F.relu(self.conv1(x, edge_index, edge_weight))
        x = F.dropout(x, p=self.dropout, training=self.training)
        x = F.dropout(self.conv2(x, edge_index, edge_weight), p=self.dropout, training=self.training)
        return F.log_softmax(self.linear(_pair_rows(x, query_edges)), dim=1)


class DiGCN_Inception_Block_link_prediction(nn.Module):
    """nn/directed/DiGCN_Inception_Block_link_prediction.py:10-80: three hidden-width inception blocks, a
    Linear head on the concatenated endpoint rows (argument order features, edge_index_tuple, query_edges,
    edge_weight_tuple)."""

    def __init__(self, num_features: int, hidden: int, label_dim: int, dropout: float = 0.5):
        super().__init__()
        self.ib1 = DiGCN_InceptionBlock(num_features, hidden)
        self.ib2 = DiGCN_InceptionBlock(hidden, hidden)
        self.ib3 = DiGCN_InceptionBlock(hidden, hidden)
        self.linear = nn.Linear(hidden * 2, label_dim)
        self._dropout = dropout
        self.reset_parameters()

    def reset_parameters(self):
        for m in (self.ib1, self.ib2, self.ib3, self.linear):
            m.reset_parameters()

    def forward(self, features, edge_index_tuple, query_edges, edge_weight_tuple):
        (ei1, ei2), (ew1, ew2) = edge_index_tuple, edge_weight_tuple
        drop = lambda t: F.dropout(t, p=self._dropout, training=self.training)  # noqa: E731
        x = features
        for depth, ib in enumerate((self.ib1, self.ib2, self.ib3)):
            x0, x1, x2 = ib(x, ei1, ew1, ei2, ew2)
            x = drop(x0) + drop(x1) + drop(x2)
            if depth < 2:
                x = drop(x)
        return F.log_softmax(self.linear(_pair_rows(x, query_edges)), dim=1)


class DGCN_link_prediction(DGCN_node_classification):
    """nn/directed/DGCN_link_prediction.py:10-96: the DGCN encoder with a Linear(6 * hidden) head on endpoint
    pairs instead of the Conv1d node head."""

    def __init__(self, num_features: int, hidden: int, label_dim: int, dropout: Optional[float] = None,
                 improved: bool = False, cached: bool = False):
        super().__init__(num_features, hidden, label_dim, dropout, improved, cached)
        del self.Conv
        self.linear = nn.Linear(hidden * 6, label_dim)

    def reset_parameters(self):
        self.lin1.reset_parameters()
        self.lin2.reset_parameters()
        nn.init.zeros_(self.bias1)
        nn.init.zeros_(self.bias2)
        self.linear.reset_parameters()

    def forward(self, x, edge_index, edge_in, edge_out, query_edges, in_w=None, out_w=None):
        x = tall_linear(x, self.lin1.weight.t())
        x = self._three(x, edge_index, edge_in, edge_out, in_w, out_w, self.bias1)
        x = tall_linear(x, self.lin2.weight.t())
        x = self._three(x, edge_index, edge_in, edge_out, in_w, out_w, self.bias2)
        x = _pair_rows(x, query_edges)
        if self.dropout > 0:          # like the reference: dropout=None (its default) raises here
            x = F.dropout(x, self.dropout, training=self.training)
        return F.log_softmax(self.linear(x), dim=1)


class SSSNET_link_prediction(SSSNET_node_clustering):
    """nn/signed/SSSNET_link_prediction.py:11-157: SSSNET's SIMPA encoder, `_W_prob` twice as tall and
    applied to concatenated endpoint embeddings; returns log-probabilities only."""

    def __init__(self, nfeat: int, hidden: int, nclass: int, dropout: float, hop: int, fill_value: float,
                 directed: bool = False, bias: bool = True):
        super().__init__(nfeat, hidden, nclass, dropout, hop, fill_value, directed, bias)
        self._W_prob = Parameter(torch.FloatTensor(2 * len(self._streams) * hidden, self._num_clusters))
        self._reset_parameters()

    def forward(self, edge_index_p, edge_weight_p, edge_index_n, edge_weight_n, features, query_edges):
        xs = [tall_linear(self._dropout(self._relu(tall_linear(features, getattr(self, f"_w_{s}0")))),
                       getattr(self, f"_w_{s}1")) for s in self._streams]
        z = self._simpa(edge_index_p, edge_weight_p, edge_index_n, edge_weight_n, *xs)
        output = tall_linear(_pair_rows(z, query_edges), self._W_prob)
        if self._bias is not None:
            output = output + self._bias
        return F.log_softmax(output, dim=1)


class SGCN(nn.Module):
    """nn/signed/SGCN.py:11-97: a stack of SGCNConv layers with tanh on stored node embeddings.
    `edge_index_s` is the reference's [E, 3] (source, target, sign) list; `forward()` takes no arguments and
    returns the embedding z; `loss()` = link-sign entropy + lamb * structure loss."""

    def __init__(self, node_num: int, edge_index_s: torch.Tensor, in_dim: int = 64, out_dim: int = 64,
                 layer_num: int = 2, init_emb: Optional[torch.Tensor] = None, init_emb_grad: bool = False,
                 lamb: float = 5, norm_emb: bool = False, **kwargs):
        super().__init__(**kwargs)
        from ..utils.signed import Link_Sign_Entropy_Loss, Sign_Structure_Loss, create_spectral_features
        self.node_num, self.in_dim, self.out_dim, self.lamb = node_num, in_dim, out_dim, lamb
        self.device = edge_index_s.device
        self.pos_edge_index = edge_index_s[edge_index_s[:, 2] > 0][:, :2].t().contiguous()
        self.neg_edge_index = edge_index_s[edge_index_s[:, 2] < 0][:, :2].t().contiguous()
        if init_emb is None:
            init_emb = create_spectral_features(self.pos_edge_index, self.neg_edge_index, node_num, in_dim
                                                ).to(self.device)
        self.x = Parameter(init_emb, requires_grad=init_emb_grad)
        self.conv1 = SGCNConv(in_dim, out_dim // 2, first_aggr=True)
        self.convs = nn.ModuleList(SGCNConv(out_dim // 2, out_dim // 2, first_aggr=False, norm_emb=norm_emb)
                                   for _ in range(layer_num - 1))
        self.lsp_loss = Link_Sign_Entropy_Loss(out_dim)
        self.structure_loss = Sign_Structure_Loss()
        self.reset_parameters()

    def reset_parameters(self):
        self.conv1.reset_parameters()
        for conv in self.convs:
            conv.reset_parameters()

    def _apply(self, fn, *args, **kwargs):          # the stored edge lists follow .to(device), like the embeddings
        out = super()._apply(fn, *args, **kwargs)
        self.pos_edge_index, self.neg_edge_index = fn(self.pos_edge_index), fn(self.neg_edge_index)
        self.device = self.x.device
        return out

    def forward(self) -> torch.Tensor:
        z = torch.tanh(self.conv1(self.x, self.pos_edge_index, self.neg_edge_index))
        for conv in self.convs:
            z = torch.tanh(conv(z, self.pos_edge_index, self.neg_edge_index))
        return z

    def loss(self) -> torch.Tensor:
        z = self.forward()
        return (self.lsp_loss(z, self.pos_edge_index, self.neg_edge_index)
                + self.lamb * self.structure_loss(z, self.pos_edge_index, self.neg_edge_index))


class SNEA(SGCN):
    """nn/signed/SNEA.py:13-93: SGCN's scaffold with SNEAConv attention layers, trainable initial embeddings by
    default, and a final tanh(Linear)."""

    def __init__(self, node_num: int, edge_index_s: torch.Tensor, in_dim: int = 64, out_dim: int = 64,
                 layer_num: int = 2, init_emb: Optional[torch.Tensor] = None, init_emb_grad: bool = True,
                 lamb: float = 4):
        from .signed.SNEAConv import SNEAConv
        super().__init__(node_num, edge_index_s, in_dim, out_dim, 1, init_emb, init_emb_grad, lamb)
        self.conv1 = SNEAConv(in_dim, out_dim // 2, first_aggr=True)
        self.convs = nn.ModuleList(SNEAConv(out_dim // 2, out_dim // 2, first_aggr=False)
                                   for _ in range(layer_num - 1))
        self.weight = nn.Linear(out_dim, out_dim)
        self.reset_parameters()

    def reset_parameters(self):
        super().reset_parameters()
        if hasattr(self, "weight"):
            self.weight.reset_parameters()

    def forward(self) -> torch.Tensor:
        return torch.tanh(self.weight(super().forward()))


def _signed_matrices(edge_index_s, n):
    """0/1 scipy CSR matrices of the positive and of the negative edges of an [E, 3] (source, target, sign)
    list; duplicate listings collapse (the reference keeps neighbour SETS)."""
    import numpy as np
    import scipy.sparse as sp
    e = edge_index_s.detach().cpu().numpy()

    def binary(rows):
        m = sp.coo_matrix((np.ones(len(rows), np.int64), (rows[:, 0], rows[:, 1])), shape=(n, n)).tocsr()
        m.sum_duplicates()
        m.data[:] = 1
        return m

    return binary(e[e[:, 2] > 0]), binary(e[e[:, 2] < 0])


def _motif_counts(P, N):
    """The 16 common-neighbour counts of the reference's `get_features` / `get_tri_features` (SDGNN.py:147-196,
    SiGAT.py:88-137) for ALL node pairs at once, in its order d1_1 .. d4_4:
    d1 = out(u) & in(v): X Y;  d2 = out(u) & out(v): X Y^T;  d3 = in(u) & out(v): X^T Y^T;  d4 = in(u) & in(v): X^T Y,
    each for (X, Y) = (P, P), (P, N), (N, P), (N, N)."""
    PT, NT = P.T.tocsr(), N.T.tocsr()
    pairs = ((P, N), (PT, NT))
    out = []
    for left, right in ((0, 0), (0, 1), (1, 1), (1, 0)):          # d1: X Y, d2: X Y^T, d3: X^T Y^T, d4: X^T Y
        for x in range(2):
            for y in range(2):
                out.append((pairs[left][x] @ pairs[right][y]).tocsr())
    return out


def _edge_list(m, device):
    """scipy matrix -> [2, E] (row node, column node) LongTensor (E may be 0)."""
    import numpy as np
    coo = m.tocoo()
    keep = coo.data != 0
    return torch.from_numpy(np.stack([coo.row[keep], coo.col[keep]]).astype(np.int64)).to(device)


class SDGNN(nn.Module):
    """nn/signed/SDGNN.py:66-267: trainable node embeddings through `layer_num` SDRLayers (four GATConv
    aggregators: positive out / in, negative out / in neighbourhoods), objective = sign + lamb_d * direction +
    lamb_t * triangle loss.  The reference enumerates neighbour sets and triangle motifs with Python dict / set
    loops; here the same sets are sparse 0/1 matrices and the 12 motif counts are sparse products (host side, once
    at construction -- not part of the device path)."""

    def __init__(self, node_num: int, edge_index_s, in_dim: int = 20, out_dim: int = 20, layer_num: int = 2,
                 init_emb: Optional[torch.Tensor] = None, init_emb_grad: bool = True, lamb_d: float = 5.0,
                 lamb_t: float = 1.0, **kwargs):
        super().__init__(**kwargs)
        from ..utils.signed import (Sign_Direction_Loss, Sign_Product_Entropy_Loss, Sign_Triangle_Loss,
                                    create_spectral_features)
        from .signed.GATConv import SDRLayer
        self.node_num, self.in_dim, self.out_dim, self.layer_num = node_num, in_dim, out_dim, layer_num
        self.device = edge_index_s.device
        self.lamb_d, self.lamb_t = lamb_d, lamb_t
        self.pos_edge_index = edge_index_s[edge_index_s[:, 2] > 0][:, :2].t().contiguous()
        self.neg_edge_index = edge_index_s[edge_index_s[:, 2] < 0][:, :2].t().contiguous()
        if init_emb is None:
            init_emb = create_spectral_features(self.pos_edge_index, self.neg_edge_index, node_num, in_dim
                                                ).to(self.device)
        self.x = Parameter(init_emb, requires_grad=init_emb_grad)
        self.edge_lists = self.build_edge_lists(edge_index_s)
        self.layers = []
        for i in range(layer_num):
            layer = SDRLayer(in_dim if i == 0 else out_dim, out_dim, edge_lists=self.edge_lists)
            self.add_module(f'SDRLayer_{i}', layer)
            self.layers.append(layer)
        self.loss_sign = Sign_Product_Entropy_Loss()
        self.loss_direction = Sign_Direction_Loss(emb_dim=out_dim)
        self.loss_tri = Sign_Triangle_Loss(emb_dim=out_dim, edge_weight=self.tri_weight)
        self.reset_parameters()

    def reset_parameters(self):
        for layer in self.layers:
            layer.reset_parameters()

    def build_edge_lists(self, edge_index_s):
        """The reference's `build_adj_lists` + `map_adj_to_edges` (SDGNN.py:139-250): neighbour SETS (duplicate
        listings collapse) as edge lists [2, E] (node, neighbour) for positive-out, positive-in, negative-out,
        negative-in, and `self.tri_weight[i, j]` = number of balanced motifs closed by the signed edge i -> j.
        With P / N the 0/1 matrices of positive / negative edges, the reference's 16 set-intersection counts are
        the entries of the products X Y, X Y^T, X^T Y^T, X^T Y (X, Y in {P, N}); its masks select
          positive edge: PP + PP^T + NN^T + N^T N^T + P^T P + N^T N
          negative edge: PN + NP + NP^T + P^T N^T + N^T P^T + P^T N
        and an edge listed with both signs keeps the negative count (the reference writes it last)."""
        P, N = _signed_matrices(edge_index_s, self.node_num)
        d = _motif_counts(P, N)
        m_pos = d[0] + d[4] + d[7] + d[11] + d[12] + d[15]        # mask [1,0,0,0, 1,0,0,1, 0,0,0,1, 1,0,0,1]
        m_neg = d[1] + d[2] + d[6] + d[9] + d[10] + d[13]         # mask [0,1,1,0, 0,0,1,0, 0,1,1,0, 0,1,0,0]
        only_pos = P - P.multiply(N)                              # edges whose last-written weight is the positive one
        self.tri_weight = (m_pos.multiply(only_pos) + m_neg.multiply(N)).tocsc()
        return [_edge_list(m, self.device) for m in (P, P.T, N, N.T)]   # out-neighbours of P^T = in-neighbours of P

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.pos_edge_index, self.neg_edge_index = fn(self.pos_edge_index), fn(self.neg_edge_index)
        moved = [fn(e) for e in self.edge_lists]
        self.edge_lists[:] = moved                                # the SDRLayers hold this same list object
        self.device = self.x.device
        return out

    def forward(self) -> torch.Tensor:
        x = self.x
        for layer in self.layers:
            x = layer(x)
        return x

    def loss(self) -> torch.Tensor:
        z = self.forward()
        return (self.loss_sign(z, self.pos_edge_index, self.neg_edge_index)
                + self.lamb_d * self.loss_direction(z, self.pos_edge_index, self.neg_edge_index)
                + self.lamb_t * self.loss_tri(z, self.pos_edge_index, self.neg_edge_index))


class SiGAT(nn.Module):
    """nn/signed/SiGAT.py:13-203: one GATConv aggregator per motif neighbourhood -- positive / negative
    (undirected, out, in) and, for the positive and for the negative out-edges, the 16 subsets closed by at
    least one triangle of each type -- concatenated with the embedding, then Linear-Tanh-Linear; objective =
    link-sign product loss.  The 38 neighbourhoods come from the same sparse products as SDGNN's weights."""

    def __init__(self, node_num: int, edge_index_s, in_dim: int = 20, out_dim: int = 20,
                 init_emb: Optional[torch.Tensor] = None, init_emb_grad: bool = True, **kwargs):
        super().__init__(**kwargs)
        from ..utils.signed import Link_Sign_Product_Loss, create_spectral_features
        from .signed.GATConv import GATConv
        self.in_dim, self.out_dim, self.node_num = in_dim, out_dim, node_num
        self.device = edge_index_s.device
        self.pos_edge_index = edge_index_s[edge_index_s[:, 2] > 0][:, :2].t().contiguous()
        self.neg_edge_index = edge_index_s[edge_index_s[:, 2] < 0][:, :2].t().contiguous()
        if init_emb is None:
            init_emb = create_spectral_features(self.pos_edge_index, self.neg_edge_index, node_num, in_dim
                                                ).to(self.device)
        self.x = Parameter(init_emb, requires_grad=init_emb_grad)
        self.edge_lists = self.build_edge_lists(edge_index_s)
        self.aggs = []
        for i in range(len(self.edge_lists)):
            self.aggs.append(GATConv(in_channels=in_dim, out_channels=out_dim))
            self.add_module('agg_{}'.format(i), self.aggs[-1])
        self.mlp_layer = nn.Sequential(nn.Linear(out_dim * (len(self.edge_lists) + 1), out_dim), nn.Tanh(),
                                       nn.Linear(out_dim, out_dim))
        self.lsp_loss = Link_Sign_Product_Loss()
        self.reset_parameters()

    def build_edge_lists(self, edge_index_s):
        P, N = _signed_matrices(edge_index_s, self.node_num)
        d = _motif_counts(P, N)

        def union(m):
            u = (m + m.T).tocsr()
            u.data[:] = 1
            return u

        mats = [union(P), P, P.T, union(N), N, N.T]
        mats += [P.multiply(c > 0) for c in d] + [N.multiply(c > 0) for c in d]
        return [_edge_list(m, self.device) for m in mats]

    def reset_parameters(self):
        for agg in self.aggs:
            agg.reset_parameters()

        def init_weights(m):
            if isinstance(m, nn.Linear):
                torch.nn.init.kaiming_normal_(m.weight)
                m.bias.data.fill_(0.01)
        self.mlp_layer.apply(init_weights)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.pos_edge_index, self.neg_edge_index = fn(self.pos_edge_index), fn(self.neg_edge_index)
        self.edge_lists[:] = [fn(e) for e in self.edge_lists]
        self.device = self.x.device
        return out

    def forward(self) -> torch.Tensor:
        neigh = [agg(self.x, edges) for edges, agg in zip(self.edge_lists, self.aggs)]
        return self.mlp_layer(torch.cat([self.x] + neigh, 1))

    def loss(self) -> torch.Tensor:
        return self.lsp_loss(self.forward(), self.pos_edge_index, self.neg_edge_index)
