"""SGCNConv -- drop-in for torch_geometric_signed_directed/nn/signed/SGCNConv.py:16: mean
aggregation over positive / negative incoming edges (value-less segment-mean HIP kernel), concat
with the node's own features, Linear."""
from typing import Tuple, Union

import torch
import torch.nn.functional as F
from torch import Tensor

from ... import _cabi, memo
from ...message_passing import MessagePassing
from ...dense import column_sums, tall_gram, tall_linear, tall_product
from ...sparse import GLOBAL_PATTERNS, spmm, spmm_rows_into


class _SgcnFn(torch.autograd.Function):
    """Both branches of one SGCNConv as ONE autograd node.  y = x W_big + bias with the columns of W_big ordered
    [own_b | own_u | a_1 | ... | a_m] (o columns each); the output is written IN PLACE, no concatenation:
        out[:, half_j] = y_own[:, half_j] + sum_j mean_in(pattern_j, a_j)        (half = 0: balanced, 1: unbalanced)
    -- the first aggregation of a half reads the own block as the SpMM's Z operand, later ones accumulate.  Backward:
    g_a_j = mean_in(pattern_j)^T g_out[:, half_j] written into ONE [N, m o] buffer, then
    dx = [g_out | g_a] W_big^T (one product over the two column segments), dW_big = x^T [g_out | g_a] (csrc/gram.hip),
    d bias = column sums of g_out.  No element-wise passes, no split / cat in either direction
    (reference order: aggregate, concatenate, Linear -- SGCNConv.py:101-126)."""

    @staticmethod
    def assemble(wb, wu, bb, bu, f, o, first_aggr):
        """[F_x, (2 + m) o] weight and bias of the ONE product: every block of lin_b / lin_u as a column block (zero blocks where
        a block reads the other half of x), columns [own_b | own_u | a_1 | ... | a_m]; biases ride on the own blocks."""
        if first_aggr:             # columns: own_b | own_u | agg_b(pos) | agg_u(neg)
            w_big = torch.cat([wb[:, f:].t(), wu[:, f:].t(), wb[:, :f].t(), wu[:, :f].t()], dim=1)
            m = 2
        else:                      # x = [lo | hi]; columns: own_b | own_u | pos_b | neg_b | pos_u | neg_u
            zeros = wb.new_zeros(f, o)
            top = torch.cat([wb[:, 2 * f:].t(), zeros, wb[:, :f].t(), zeros, zeros, wu[:, f:2 * f].t()], dim=1)
            bot = torch.cat([zeros, wu[:, 2 * f:].t(), zeros, wb[:, f:2 * f].t(), wu[:, :f].t(), zeros], dim=1)
            w_big = torch.cat([top, bot], dim=0)
            m = 4
        bias = None if bb is None else torch.cat([bb, bu, bb.new_zeros(m * o)])
        return w_big, bias

    @staticmethod
    def split_weight_gradient(dw, f, o, first_aggr):
        """dW of the one product -> (d lin_b.weight, d lin_u.weight): the blocks `assemble` placed, transposed back; the
        zero blocks' gradients are dropped (they are constants)."""
        if first_aggr:             # dw: [f, own_b | own_u | agg_b | agg_u]
            return (torch.cat([dw[:, 2 * o:3 * o].t(), dw[:, :o].t()], dim=1),
                    torch.cat([dw[:, 3 * o:4 * o].t(), dw[:, o:2 * o].t()], dim=1))
        lo, hi = dw[:f], dw[f:]    # rows of x = [lo | hi]; columns own_b | own_u | pos_b | neg_b | pos_u | neg_u
        return (torch.cat([lo[:, 2 * o:3 * o].t(), hi[:, 3 * o:4 * o].t(), lo[:, :o].t()], dim=1),
                torch.cat([hi[:, 4 * o:5 * o].t(), lo[:, 5 * o:6 * o].t(), hi[:, o:2 * o].t()], dim=1))

    @staticmethod
    def forward(ctx, x, wb, wu, bb, bu, f, o, first_aggr, spec):
        # Round 5: the node takes the layer's OWN parameters.  Composing [own | aggregated] blocks from slices of lin_b / lin_u
        # under autograd cost, per step, four slice-backward fills + copies and two accumulations per weight, and as many small
        # launches again for the bias -- a quarter of the C3 step was such glue (profiles/r4j_configs.json: 265 of 1039 us).
        w_big, bias = _SgcnFn.assemble(wb.detach(), wu.detach(), None if bb is None else bb.detach(),
                                       None if bu is None else bu.detach(), f, o, first_aggr)
        n = x.size(0)
        # the own block [own_b | own_u] and one contiguous matrix per aggregated block (gathered by whole rows)
        parts = tall_product([x], w_big, False, bias, splits=(2 * o,) + (o,) * len(spec))
        own = parts[0]
        out = torch.empty((n, 2 * o), dtype=x.dtype, device=x.device)
        seen = set()
        for j, (pat, half) in enumerate(spec):
            dst = out[:, half * o:(half + 1) * o]
            if half in seen:
                spmm_rows_into(pat.fwd, None, parts[1 + j], dst, accumulate=True, mean=True)
            else:
                spmm_rows_into(pat.fwd, None, parts[1 + j], dst, mean=True, z=own[:, half * o:(half + 1) * o])
                seen.add(half)
        ctx.save_for_backward(x, w_big)
        ctx.o, ctx.f, ctx.first_aggr, ctx.spec, ctx.has_bias = o, f, first_aggr, spec, bias is not None
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, w_big = ctx.saved_tensors
        o, spec = ctx.o, ctx.spec
        g = g.contiguous()
        ga = torch.empty((x.size(0), len(spec) * o), dtype=x.dtype, device=x.device)
        for j, (pat, half) in enumerate(spec):
            # backward of the mean: every entry weighs 1 / in-degree of its target (cached per pattern)
            spmm_rows_into(pat.bwd, pat.values_for(pat.mean_values(), "bwd"), g[:, half * o:(half + 1) * o],
                           ga[:, j * o:(j + 1) * o])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = tall_product([g, ga], w_big, True)          # g W_own^T + g_a W_agg^T in one pass
        d_wb = d_wu = d_bb = d_bu = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw = tall_gram([x], [g, ga])                     # [x^T g_out | x^T g_a] in one pass over x, g and g_a
            d_wb, d_wu = _SgcnFn.split_weight_gradient(dw, ctx.f, o, ctx.first_aggr)
        if ctx.has_bias and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]):
            sums = column_sums(g)                            # [2 o]: the balanced half's bias, then the unbalanced half's
            d_bb, d_bu = sums[:o], sums[o:]
        return dx, d_wb, d_wu, d_bb, d_bu, None, None, None, None


class SGCNConv(MessagePassing):
    edge_weight_arg = None
    _fused_message = True

    def __init__(self, in_dim: int, out_dim: int, first_aggr: bool, bias: bool = True,
                 norm_emb: bool = False, **kwargs):
        kwargs.setdefault('aggr', 'mean')
        super().__init__(**kwargs)
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.first_aggr = first_aggr
        self.norm_emb = norm_emb
        k = 2 if first_aggr else 3
        self.lin_b = torch.nn.Linear(k * in_dim, out_dim, bias)
        self.lin_u = torch.nn.Linear(k * in_dim, out_dim, bias)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin_b.reset_parameters()
        self.lin_u.reset_parameters()

    def _mean_in(self, x_src: Tensor, n_dst: int, edge_index: Tensor) -> Tensor:
        pat = GLOBAL_PATTERNS.get(edge_index, x_src.size(-2), n_dst, self.flow)
        return spmm(pat, x_src, None, reduce=self.aggr)

    def _branch(self, lin, aggregated, own, n):
        """lin(cat([mean_in(x_1), ..., own])).  The Linear is applied block-wise; when it narrows the
        features (in_dim > out_dim) each block is multiplied BEFORE its mean aggregation --
        mean_in(x) W = mean_in(x W) -- so the segment-mean SpMM runs at width out_dim
        (reference order: aggregate, concatenate, then Linear; SGCNConv.py:101-119)."""
        f = self.in_dim
        w = lin.weight                                  # [out_dim, (len(aggregated) + 1) * in_dim]
        if own.dim() == 2:
            lin_fn = tall_linear
        else:                                           # [..., N, F] batches: the Linear is shared, so the batch is one tall product
            def lin_fn(t, wt, b=None):
                return tall_linear(t.reshape(-1, t.size(-1)), wt, b).view(*t.shape[:-1], wt.size(1))
        if self.in_dim > self.out_dim:
            out = lin_fn(own, w[:, len(aggregated) * f:].t(), lin.bias)
            for k, (feat, ei) in enumerate(aggregated):
                out = out + self._mean_in(lin_fn(feat, w[:, k * f:(k + 1) * f].t()), n, ei)
            return out
        parts = [self._mean_in(feat, n, ei) for feat, ei in aggregated] + [own]
        return lin_fn(torch.cat(parts, dim=-1), w.t(), lin.bias)

    def _fused(self, x: Tensor, pos_edge_index: Tensor, neg_edge_index: Tensor) -> Tensor:
        """Both branches from ONE GEMM when the Linear does not widen (in_dim >= out_dim): every block of
        lin_b / lin_u becomes a column block of one [F_x, 4 or 6 * out_dim] matrix (zero blocks where a block
        reads the other half of x), the biases ride on the own-feature blocks, and the mean aggregations add their
        results onto the own blocks inside the SpMM's epilogue, written straight into the two halves of the output
        (`_SgcnFn`: one autograd node, no concatenation, no element-wise passes, one split-K weight gradient)."""
        f, o, n = self.in_dim, self.out_dim, x.size(0)
        wb, wu = self.lin_b.weight, self.lin_u.weight
        pos = GLOBAL_PATTERNS.get(pos_edge_index, n, n, self.flow)
        neg = GLOBAL_PATTERNS.get(neg_edge_index, n, n, self.flow)
        spec = ((pos, 0), (neg, 1)) if self.first_aggr else ((pos, 0), (neg, 0), (pos, 1), (neg, 1))
        if o % 4 == 0 and x.dtype == torch.float32 and self.aggr == "mean":
            return _SgcnFn.apply(x, wb, wu, self.lin_b.bias, self.lin_u.bias, f, o, self.first_aggr, spec)
        w_big, bias = _SgcnFn.assemble(wb, wu, self.lin_b.bias, self.lin_u.bias, f, o, self.first_aggr)   # (under autograd)
        y = tall_linear(x, w_big, bias).split(o, dim=1)    # widths the 16-byte-row kernels do not slice: composed ops
        halves = [y[0], y[1]]
        for j, (pat, half) in enumerate(spec):
            halves[half] = spmm(pat, y[2 + j], None, z=halves[half], beta=1.0, reduce=self.aggr)
        return torch.cat(halves, dim=-1)

    def forward(self, x: Union[Tensor, Tuple[Tensor, Tensor]], pos_edge_index: Tensor,
                neg_edge_index: Tensor) -> Tensor:
        # (one content check of the two edge lists for all pattern lookups of this forward: memo.verified)
        with memo.verified(pos_edge_index if isinstance(pos_edge_index, Tensor) else None,
                           neg_edge_index if isinstance(neg_edge_index, Tensor) else None):
            return self._forward(x, pos_edge_index, neg_edge_index)

    def _forward(self, x, pos_edge_index, neg_edge_index):
        if (isinstance(x, Tensor) and x.dim() == 2 and self.in_dim >= self.out_dim
                and isinstance(pos_edge_index, Tensor) and isinstance(neg_edge_index, Tensor)):
            _cabi.require_gpu(x, pos_edge_index, neg_edge_index)
            out = self._fused(x, pos_edge_index, neg_edge_index)
            return F.normalize(out, p=2, dim=-1) if self.norm_emb else out
        if isinstance(x, Tensor):
            x = (x, x)
        if not isinstance(pos_edge_index, Tensor) or not isinstance(neg_edge_index, Tensor):
            raise NotImplementedError("SGCNConv: only the edge_index (Tensor) path exists on the HIP stack")
        _cabi.require_gpu(x[0], x[1], pos_edge_index, neg_edge_index)
        n = x[1].size(-2)
        if self.first_aggr:
            out_b = self._branch(self.lin_b, [(x[0], pos_edge_index)], x[1], n)
            out_u = self._branch(self.lin_u, [(x[0], neg_edge_index)], x[1], n)
        else:
            f = self.in_dim
            lo, hi = x[0][..., :f], x[0][..., f:]   # column slices: passed by row stride, no copy
            out_b = self._branch(self.lin_b, [(lo, pos_edge_index), (hi, neg_edge_index)], x[1][..., :f], n)
            out_u = self._branch(self.lin_u, [(hi, pos_edge_index), (lo, neg_edge_index)], x[1][..., f:], n)
        out = torch.cat([out_b, out_u], dim=-1)
        if self.norm_emb:
            out = F.normalize(out, p=2, dim=-1)
        return out

    def message(self, x_j: Tensor) -> Tensor:
        return x_j

    def __repr__(self) -> str:
        return (f'{self.__class__.__name__}({self.in_dim}, '
                f'{self.out_dim}, first_aggr={self.first_aggr})')
