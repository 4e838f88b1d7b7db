"""SIMPA -- drop-in for torch_geometric_signed_directed/nn/signed/SIMPA.py:11 (signed mixed-path
aggregation of SSSNET): a hop schedule of Conv_Base SpMMs and axpys."""
from typing import Optional

import torch
from torch.nn import Parameter

from ... import _cabi
from ... import memo
from ...memo import TensorMemo
from ...sparse import GLOBAL_PATTERNS, _spmm_raw
from ..general.conv_base import Conv_Base, flipped_edge_index


def weighted_sum(tensors, weights, out=None):
    """sum_j weights[j] * tensors[j] (fp32 matrices of one shape, Python floats) in ONE pass: pygsd_weighted_sum_f32.
    out: where to write it -- a column block of a wider matrix qualifies (unit column stride, 16-byte aligned rows)."""
    import ctypes
    k = len(tensors)
    tensors = [_cabi.c16(t) for t in tensors]           # (a product at an odd width is a column slice of a padded one)
    shape = tensors[0].shape
    rows, cols = (1, tensors[0].numel()) if tensors[0].dim() != 2 else (shape[0], shape[1])
    if out is None:
        out = torch.empty_like(tensors[0])
        ldo = cols
    else:
        if tuple(out.shape) != tuple(shape) or out.dim() != 2 or out.stride(1) != 1 or out.dtype != torch.float32:
            raise ValueError("weighted_sum: `out` must be an fp32 [rows, cols] matrix with unit column stride")
        ldo = out.stride(0)
    ptrs = (ctypes.c_void_p * k)(*[t.data_ptr() for t in tensors])
    ws = (ctypes.c_float * k)(*[float(w) for w in weights])
    with _cabi.on_device(out.device):
        _cabi.check(_cabi.lib().pygsd_weighted_sum_f32(ptrs, ws, k, rows, cols, _cabi.ptr(out), ldo, _cabi.stream_ptr()),
                    "pygsd_weighted_sum_f32")
    return out


def dots(g, tensors):
    """[<g, t> for t in tensors] as one fp32 device vector, g read once (pygsd_dots_f32); at most 8 tensors of g's shape."""
    import ctypes
    k = len(tensors)
    g = _cabi.c16(g)
    tensors = [_cabi.c16(t) for t in tensors]
    rows, cols = (1, g.numel()) if g.dim() != 2 else (g.size(0), g.size(1))
    out = torch.empty(k, dtype=torch.float32, device=g.device)
    ws = torch.empty(8192, dtype=torch.float64, device=g.device)      # 64 KiB: 1024 blocks x 8 products, float64 partials
    ptrs = (ctypes.c_void_p * k)(*[t.data_ptr() for t in tensors])
    with _cabi.on_device(g.device):
        _cabi.check(_cabi.lib().pygsd_dots_f32(_cabi.ptr(g), cols, ptrs, k, rows, cols, _cabi.ptr(out), _cabi.ptr(ws),
                                               ws.numel() * 8, _cabi.stream_ptr()), "pygsd_dots_f32")
    return out


_HOST_WEIGHTS = TensorMemo(16, verify=False)      # (its own contract: see _host_weights)


def _host_weights(w):
    """The hop weights as Python floats (they become alpha / beta kernel arguments).

    While the weights are being TRAINED (grad mode on, `requires_grad`) the live tensor is read on every call, exactly as the
    reference does (SIMPA.py:77-93 multiplies by the parameter itself): an optimiser step changes it between forwards anyway, and
    the idioms that write a trainable parameter behind the version counter (`p.data.clamp_()`, `p.data.copy_()`) must not meet an
    old copy.  Only inference (`torch.no_grad()` / frozen weights) re-uses one device -> host read per in-place VERSION of the
    parameter (memo.TensorMemo: weakly held, same opt-outs): no synchronisation in a forward whose weights did not change."""
    if torch.is_grad_enabled() and w.requires_grad:
        return w.detach().reshape(-1).tolist()
    hit = _HOST_WEIGHTS.get((w,), "hop weights")
    if hit is None:
        hit = _HOST_WEIGHTS.put((w,), "hop weights", tuple(w.detach().reshape(-1).tolist()))
    return list(hit)


class _StreamFn(torch.autograd.Function):
    """One (positive, negative) stream of SIMPA as ONE autograd node (fixed operator values).

    Forward: the hop schedule of SIMPA.py:77-93 -- every node of it is a HIP SpMM of an earlier one; feat_p / feat_n are
    weighted sums of nodes.  Backward: the gradient of a node is  sum_terms w g_feat + sum_consumers S^T grad_consumer ;
    both kinds of summand ride on the SpMM's own epilogue (Y = alpha S^T X + beta Z: a pending `w g` is consumed as
    (x = g, alpha = w) when it is propagated and as (z = g, beta = w) when something is added to it), so the chain
    costs its SpMMs and nothing else -- autograd's composition paid a scale pass, a product-and-reduce pass and an
    accumulation pass per hop on top.  The hop weights' gradients are dot products <g_feat, node>, one pass per stream
    half (pygsd_dots_f32).  feat_p / feat_n are written side by side into ONE [N, 2F] matrix (the reference's cat,
    SIMPA.py:95).  The weights are read to the host once per parameter version (`_host_weights`; six floats at hop 2)."""

    @staticmethod
    def forward(ctx, x_pos, x_neg, wp, wn, op_p, op_n, hop):
        wpl, wnl = _host_weights(wp), _host_weights(wn)
        nodes, ops, terms_p, terms_n = [x_pos.contiguous(), x_neg.contiguous()], [], [(0, 0)], []

        def apply(kind, src):
            pat, w = op_p if kind == "p" else op_n
            nodes.append(_spmm_raw(pat.fwd, pat.values_for(w, "fwd"), nodes[src], None, 1.0, 0.0, False))
            ops.append((len(nodes) - 1, kind, src))
            return len(nodes) - 1

        cur_p, aux_n, j = 0, 1, 0
        for h in range(hop + 1):
            if h > 0:
                cur_p = apply("p", cur_p)
                if h != hop:           # the reference also advances aux_n at the last hop, but never reads it again
                    aux_n = apply("p", aux_n)
                terms_p.append((h, cur_p))
            if h != hop:
                cur_n = apply("n", aux_n)
                terms_n.append((j, cur_n))
                j += 1
                for _ in range(hop - 1 - h):
                    cur_n = apply("p", cur_n)
                    terms_n.append((j, cur_n))
                    j += 1

        n, f = nodes[0].shape
        feat = torch.empty((n, 2 * f), dtype=nodes[0].dtype, device=nodes[0].device)   # [feat_p | feat_n], SIMPA.py:95

        def weighted(terms, weights, out):
            if not terms:
                out.zero_()
            elif f % 4 == 0 and len(terms) <= 8:
                weighted_sum([nodes[v] for _, v in terms], [weights[wi] for wi, _ in terms], out)
            else:
                torch.mul(nodes[terms[0][1]], weights[terms[0][0]], out=out)
                for wi, v in terms[1:]:
                    out.add_(nodes[v], alpha=weights[wi])

        weighted(terms_p, wpl, feat[:, :f])
        weighted(terms_n, wnl, feat[:, f:])
        ctx.save_for_backward(*nodes)
        ctx.tape = (ops, terms_p, terms_n, wpl, wnl, op_p, op_n, tuple(wp.shape), tuple(wn.shape))
        return feat

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        nodes = ctx.saved_tensors
        ops, terms_p, terms_n, wpl, wnl, op_p, op_n, shape_p, shape_n = ctx.tape
        f = nodes[0].size(1)
        # contiguous halves: most of the backward products gather their rows straight from these
        g_p, g_n = g[:, :f].contiguous(), g[:, f:].contiguous()
        grads = {}                                   # node -> tensor | ("pending", w, g): w * g, not materialised

        def add_term(v, w, g):
            cur = grads.get(v)
            if cur is None:
                grads[v] = ("pending", w, g)
            elif isinstance(cur, tuple):
                grads[v] = ("pending", cur[1] + w, g) if cur[2] is g else cur[2] * cur[1] + g * w
            else:
                cur.add_(g, alpha=w)

        for wi, v in terms_p:
            add_term(v, wpl[wi], g_p)
        for wi, v in terms_n:
            add_term(v, wnl[wi], g_n)
        need = ctx.needs_input_grad
        chain = ops if (need[0] or need[1]) else ()   # frozen inputs (a first layer on fixed features): no product at all
        for dst, kind, src in reversed(chain):       # every consumer of `dst` was created later: its gradient is complete
            gd = grads.pop(dst, None)
            if gd is None:
                continue
            pat, w = op_p if kind == "p" else op_n
            x, alpha = (gd[2], gd[1]) if isinstance(gd, tuple) else (gd, 1.0)
            cs = grads.get(src)
            if cs is None:
                z, beta = None, 0.0
            elif isinstance(cs, tuple):
                z, beta = cs[2], cs[1]
            else:
                z, beta = cs, 1.0
            grads[src] = _spmm_raw(pat.bwd, pat.values_for(w, "bwd"), x, z, alpha, beta, False)

        def materialise(v):
            cur = grads.get(v)
            if cur is None:
                return torch.zeros_like(nodes[v])
            return cur[2] * cur[1] if isinstance(cur, tuple) else cur

        def weight_grads(terms, g, shape):
            out = g.new_zeros(shape)
            if not terms:
                return out
            flat = out.view(-1)
            if g.size(1) % 4 == 0 and len(terms) <= 8:
                res = dots(g, [nodes[v] for _, v in terms])             # g read once for all of the stream's hop weights
                which = [wi for wi, _ in terms]
                if which == list(range(flat.numel())):
                    return res.view(shape)
                flat.index_copy_(0, torch.tensor(which, device=g.device), res)
            else:
                gd = g.reshape(-1).double()                              # widths the kernel does not vectorise: the same float64 sums
                for wi, v in terms:
                    flat[wi] = torch.dot(gd, nodes[v].reshape(-1).double())
            return out

        return (materialise(0) if need[0] else None, materialise(1) if need[1] else None,
                weight_grads(terms_p, g_p, shape_p) if need[2] else None,
                weight_grads(terms_n, g_n, shape_n) if need[3] else None, None, None, None)


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
        self._fill_weights(("_w_p", "_w_n"))

    def _reset_parameters_directed(self):
        self._fill_weights(("_w_sp", "_w_sn", "_w_tp", "_w_tn"))

    def _fill_weights(self, names):
        # an in-place write under no_grad bumps the version counter (`.data.fill_` would not), and the host copies of the hop
        # weights are dropped outright: a reset can never meet the scalars of the old values
        with torch.no_grad():
            for name in names:
                getattr(self, name).fill_(1.0)
        _HOST_WEIGHTS.clear()

    def _fusable(self, w_p, w_n, x_pos, x_neg):
        return (x_pos.dim() == 2 and x_pos.is_cuda and x_pos.dtype == torch.float32 and x_neg.dtype == torch.float32
                and x_pos.shape == x_neg.shape and not (w_p is not None and w_p.requires_grad)
                and not (w_n is not None and w_n.requires_grad)
                and all(c.normalize for c in (self.conv_layer_p, self.conv_layer_n)))

    def _stream(self, ei_p, w_p, ei_n, w_n, x_pos, x_neg, wp, wn):
        """One (positive, negative) feature pair: feat_p = sum_h wp[h] Ap^h x_pos and the mixed
        paths Ap^m An Ap^h x_neg, in the reference's accumulation order (SIMPA.py:77-93).  -> [feat_p | feat_n]."""
        if self._fusable(w_p, w_n, x_pos, x_neg):
            _cabi.require_gpu(x_pos, x_neg, ei_p, ei_n, w_p, w_n)
            n = x_pos.size(0)
            handles = []
            for conv, ei, w in ((self.conv_layer_p, ei_p, w_p), (self.conv_layer_n, ei_n, w_n)):
                nei, nw = conv._normalised(ei, w, n, x_pos.dtype)          # conv_norm_rw, memoised on the graph tensors
                handles.append((GLOBAL_PATTERNS.get(nei, n, n, conv.flow, validate=False, trusted=True), nw))
            return _StreamFn.apply(x_pos, x_neg, wp, wn, handles[0], handles[1], self._hop_p - 1)    # [feat_p | feat_n]
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
        return torch.cat([feat_p, torch.zeros_like(feat_p) if feat_n is None else feat_n], dim=1)

    def forward(self, edge_index_p: torch.LongTensor, edge_weight_p: torch.FloatTensor,
                edge_index_n: torch.LongTensor, edge_weight_n: torch.FloatTensor,
                x_p: torch.FloatTensor, x_n: torch.FloatTensor,
                x_pt: Optional[torch.FloatTensor] = None,
                x_nt: Optional[torch.FloatTensor] = None) -> torch.FloatTensor:
        # one content check of the four graph tensors for every memo lookup of this forward (memo.verified); the flipped lists
        # are this package's own tensors, derived inside the scope
        with memo.verified(edge_index_p, edge_weight_p, edge_index_n, edge_weight_n):
            if self._undirected:
                return self._stream(edge_index_p, edge_weight_p, edge_index_n, edge_weight_n, x_p, x_n, self._w_p, self._w_n)
            source = self._stream(edge_index_p, edge_weight_p, edge_index_n, edge_weight_n, x_p, x_n, self._w_sp, self._w_sn)
            flip_p, flip_n = flipped_edge_index(edge_index_p), flipped_edge_index(edge_index_n)
            memo.trust(flip_p, flip_n)
            target = self._stream(flip_p, edge_weight_p, flip_n, edge_weight_n, x_pt, x_nt, self._w_tp, self._w_tn)
        return torch.cat([source, target], dim=1)               # [sp | sn | tp | tn], SIMPA.py:142
