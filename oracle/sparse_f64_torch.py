"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The formulas of oracle/sparse_f64.py (float64 evaluation of the MagNetConv / MSConv operator and layer, any K, with
all gradients) and of the DiGCN propagate, restated on torch tensors so that they run on WHATEVER device the inputs
live on -- the host for the pinning test (tests/test_oracle_sparse_f64.py holds this file to the scipy evaluation,
<= 1e-12), the MI355X itself for the checks at the BASELINE configs' stated sizes (1M nodes / 20M edges at h = 128,
K = 2; 2M nodes / 52M entries), where the scipy evaluation needs minutes of one host core per configuration.

It shares nothing with the product: float64 throughout, duplicates summed by `torch.sparse_coo_tensor(...).coalesce()`,
products as chunked `index_add_` over COO entries (no CSR, no HIP kernel of this repository, no torch.sparse matmul).
Reference formulas: utils/directed/get_magnetic_Laplacian.py:10-93, utils/general/get_magnetic_signed_Laplacian.py:10-98,
nn/directed/MagNetConv.py:185-249 (out_real = A - B + b, out_imag = A + B + b with A = cheb(Re S^T, X_r),
B = cheb(Im S^T, X_i)), nn/directed/DiGCNConv.py:54-94 (out = S^T (x W) + b, aggregation at edge_index[1]).
"""
import math

import torch

_CHUNK = 1 << 22          # COO entries per index_add_ (bounds the [chunk, F] float64 temporary)


def spmm(rows, cols, vals, x, n_rows):
    """out[rows[e]] += vals[e] * x[cols[e]] in float64 (vals None = ones)."""
    out = torch.zeros((n_rows, x.size(1)), dtype=torch.float64, device=x.device)
    for lo in range(0, rows.numel(), _CHUNK):
        sl = slice(lo, lo + _CHUNK)
        msg = x.index_select(0, cols[sl])
        if vals is not None:
            msg = msg * vals[sl].unsqueeze(1)
        out.index_add_(0, rows[sl], msg)
    return out


class Operator:
    """S = 2 L / lambda_max - I as COO (float64): off-diagonal entries (row, col, real, imag) + the diagonal."""

    def __init__(self, row, col, real, imag, diag, n):
        self.row, self.col, self.real, self.imag, self.diag, self.n = row, col, real, imag, diag, n

    def apply_t(self, part, x):
        """(Re or Im S)^T x: the propagate aggregates at the COLUMN (target) index; the diagonal is real."""
        vals = self.real if part == "real" else self.imag
        y = spmm(self.col, self.row, vals, x, self.n)
        return y + self.diag.unsqueeze(1) * x if part == "real" else y

    def apply(self, part, x):
        vals = self.real if part == "real" else self.imag
        y = spmm(self.row, self.col, vals, x, self.n)
        return y + self.diag.unsqueeze(1) * x if part == "real" else y


def magnetic_operator(edge_index, edge_weight, n, q, normalization="sym", lambda_max=2.0, signed=False,
                      absolute_degree=True):
    ei = edge_index
    dev = ei.device
    w = torch.ones(ei.size(1), dtype=torch.float64, device=dev) if edge_weight is None else edge_weight.double()
    keep = ei[0] != ei[1]
    u, v, w = ei[0][keep], ei[1][keep], w[keep]
    idx = torch.cat([torch.stack([u, v]), torch.stack([v, u])], dim=1)

    def summed(values):                 # duplicates summed; coalesce orders by (row, col): the patterns line up
        return torch.sparse_coo_tensor(idx, values, (n, n)).coalesce()

    a_s = summed(torch.cat([0.5 * w, 0.5 * w]))
    arg = summed(torch.cat([w, -w])).values()
    row, col = a_s.indices()
    mag = a_s.values()
    zeros = torch.zeros(n, dtype=torch.float64, device=dev)
    if not signed:
        d = zeros.index_add(0, row, mag)
    elif absolute_degree:
        d = zeros.index_add(0, row, summed(torch.cat([0.5 * w.abs(), 0.5 * w.abs()])).values())
    else:
        d = zeros.index_add(0, row, mag.abs())
    if normalization == "sym":
        dis = torch.where(d != 0, d.pow(-0.5), torch.zeros_like(d))
        mag = dis[row] * mag * dis[col]
        diag = torch.ones(n, dtype=torch.float64, device=dev)
    else:
        diag = d
    phase = (2.0 * math.pi * q) * arg
    scale = 2.0 / lambda_max
    # L = diag - H  ->  S = scale * L - I
    return Operator(row, col, -scale * mag * torch.cos(phase), -scale * mag * torch.sin(phase), scale * diag - 1.0, n)


def _cheb_terms(op, part, x, k1):
    ts = [x]
    if k1 > 1:
        ts.append(op.apply_t(part, x))
    for _ in range(2, k1):
        ts.append(2.0 * op.apply_t(part, ts[-1]) - ts[-2])
    return ts


def magnet_conv(x_real, x_imag, op, weight, bias, g_real=None, g_imag=None):
    """(out_real, out_imag) and, with upstream gradients, also (dx_real, dx_imag, dweight, dbias) of
    <out_real, g_real> + <out_imag, g_imag> -- the same contract as oracle/sparse_f64.magnet_conv."""
    xr, xi, w = x_real.double(), x_imag.double(), weight.double()
    k1 = w.size(0)
    ta, tb = _cheb_terms(op, "real", xr, k1), _cheb_terms(op, "imag", xi, k1)
    a = sum(ta[k] @ w[k] for k in range(k1))
    b = sum(tb[k] @ w[k] for k in range(k1))
    bb = 0.0 if bias is None else bias.double()
    out = (a - b + bb, a + b + bb)
    if g_real is None:
        return out
    gr, gi = g_real.double(), g_imag.double()
    p, mm = gr + gi, gi - gr
    dw = torch.stack([ta[k].t() @ p + tb[k].t() @ mm for k in range(k1)])
    db = p.sum(0)

    def adjoint(part, up):              # Clenshaw with the un-transposed operator on up W_k^T
        c = [up @ w[k].t() for k in range(k1)]
        b1 = torch.zeros_like(c[0])
        b2 = torch.zeros_like(c[0])
        for k in range(k1 - 1, 0, -1):
            b1, b2 = c[k] + 2.0 * op.apply(part, b1) - b2, b1
        return c[0] + op.apply(part, b1) - b2

    return out + (adjoint("real", p), adjoint("imag", mm), dw, db)


def digcn_conv(x, edge_index, edge_weight, weight, bias, g=None):
    """out = S^T (x W) + b with out[target] += w * (x W)[source]; with an upstream gradient also (dx, dW, db)."""
    x, w, ew = x.double(), weight.double(), edge_weight.double()
    h = x @ w
    out = spmm(edge_index[1], edge_index[0], ew, h, x.size(0))
    if bias is not None:
        out = out + bias.double()
    if g is None:
        return out
    g = g.double()
    dh = spmm(edge_index[0], edge_index[1], ew, g, x.size(0))
    return out, dh @ w.t(), x.t() @ dh, g.sum(0)


# --------------------------------------------------------------------------------------------------
# Signed scatter-aggregate layers (BASELINE config C3: SSSNET / SGCN on an SSBM graph) in float64.
# Reference formulas: nn/general/conv_base.py:12-31 (conv_norm_rw: add_remaining_self_loops, row sums, D^-1) and
# :98-114 (flow = target_to_source: aggregation at edge_index[0]); nn/signed/SIMPA.py:77-139 (hop schedule);
# nn/directed/DIMPA.py:32-59; nn/signed/SGCNConv.py:94-126 (mean over incoming edges, cat, Linear);
# nn/signed/SSSNET_node_clustering.py:90-160 (two-layer MLPs -> SIMPA -> linear head, softmax, normalize).
# Gradients come from torch autograd through `_Apply` (a fixed linear operator and its transpose), in float64.
# --------------------------------------------------------------------------------------------------
class _Apply(torch.autograd.Function):
    """y = A x for a fixed COO operator A (rows, cols, vals): chunked index_add_, adjoint = A^T g."""

    @staticmethod
    def forward(ctx, x, rows, cols, vals, n_rows):
        ctx.op = (rows, cols, vals, x.size(0))
        return spmm(rows, cols, vals, x, n_rows)

    @staticmethod
    def backward(ctx, g):
        rows, cols, vals, n_cols = ctx.op
        return spmm(cols, rows, vals, g.contiguous(), n_cols), None, None, None, None


class RowOperator:
    """out[rows[e]] += vals[e] * x[cols[e]] (float64), differentiable in x."""

    def __init__(self, rows, cols, vals, n):
        self.rows, self.cols, self.vals, self.n = rows, cols, vals, n

    def __call__(self, x):
        return _Apply.apply(x, self.rows, self.cols, self.vals, self.n)


def rw_operator(edge_index, edge_weight, n, fill, flip=False):
    """D^-1 (A - listed loops + diag(loop weight if listed (last one wins) else fill)), aggregated at edge_index[0]
    (conv_base.py:12-31, :98-114).  flip: on edge_index[[1, 0]] (the target streams of directed SIMPA / DIMPA)."""
    ei = edge_index[[1, 0]] if flip else edge_index
    dev = ei.device
    w = torch.ones(ei.size(1), dtype=torch.float64, device=dev) if edge_weight is None else edge_weight.double()
    off = ei[0] != ei[1]
    diag = torch.full((n,), float(fill), dtype=torch.float64, device=dev)
    pos = (~off).nonzero(as_tuple=True)[0]
    if pos.numel():
        last = torch.full((n,), -1, dtype=torch.long, device=dev).scatter_reduce(0, ei[0][pos], pos, "amax",
                                                                                 include_self=True)
        diag = torch.where(last >= 0, w[last.clamp(min=0)], diag)
    loops = torch.arange(n, device=dev)
    rows, cols, vals = torch.cat([ei[0][off], loops]), torch.cat([ei[1][off], loops]), torch.cat([w[off], diag])
    deg = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, rows, vals)
    inv = torch.where(deg != 0, 1.0 / deg, torch.zeros_like(deg))
    return RowOperator(rows, cols, inv[rows] * vals, n)


def mean_operator(edge_index, n):
    """Row i = mean of x[j] over the listed edges j -> i (a multi-edge counts twice; no edge: 0)."""
    src, dst = edge_index[0], edge_index[1]
    cnt = torch.zeros(n, dtype=torch.float64, device=src.device).index_add_(
        0, dst, torch.ones(dst.numel(), dtype=torch.float64, device=src.device))
    return RowOperator(dst, src, (1.0 / cnt.clamp(min=1.0))[dst], n)


def simpa_stream(a_p, a_n, x_pos, x_neg, wp, wn, hop):
    """(feat_p, feat_n): feat_p = sum_h wp[h] Ap^h x_pos; feat_n = sum_{h < hop} sum_{m < hop - h} wn[j] Ap^m An Ap^h x_neg,
    j counting h-major (SIMPA.py:77-93)."""
    wp, wn = wp.reshape(-1), wn.reshape(-1)
    feat_p = wp[0] * x_pos
    cur = x_pos
    for h in range(1, hop + 1):
        cur = a_p(cur)
        feat_p = feat_p + wp[h] * cur
    feat_n = torch.zeros_like(feat_p)
    base, j = x_neg, 0
    for h in range(hop):
        cur = a_n(base)
        for m in range(hop - h):
            if m:
                cur = a_p(cur)
            feat_n = feat_n + wn[j] * cur
            j += 1
        base = a_p(base)
    return feat_p, feat_n


def simpa(ei_p, w_p, ei_n, w_n, x_p, x_n, params, hop, fill, directed=False, x_pt=None, x_nt=None):
    """SIMPA.forward in float64 (params: the module's weights as float64 tensors, possibly requiring grad)."""
    n = x_p.size(0)
    a_p, a_n = rw_operator(ei_p, w_p, n, fill), rw_operator(ei_n, w_n, n, 0.0)
    if not directed:
        return torch.cat(simpa_stream(a_p, a_n, x_p, x_n, params["_w_p"], params["_w_n"], hop), dim=1)
    t_p, t_n = rw_operator(ei_p, w_p, n, fill, flip=True), rw_operator(ei_n, w_n, n, 0.0, flip=True)
    return torch.cat(simpa_stream(a_p, a_n, x_p, x_n, params["_w_sp"], params["_w_sn"], hop)
                     + simpa_stream(t_p, t_n, x_pt, x_nt, params["_w_tp"], params["_w_tn"], hop), dim=1)


def dimpa(x_s, x_t, edge_index, edge_weight, w_s, w_t, hop, fill=0.5):
    """DIMPA.forward in float64 (DIMPA.py:32-59)."""
    n = x_s.size(0)
    a, a_t = rw_operator(edge_index, edge_weight, n, fill), rw_operator(edge_index, edge_weight, n, fill, flip=True)
    w_s, w_t = w_s.reshape(-1), w_t.reshape(-1)
    feat_s, feat_t, cur_s, cur_t = w_s[0] * x_s, w_t[0] * x_t, x_s, x_t
    for h in range(1, hop + 1):
        cur_s, cur_t = a(cur_s), a_t(cur_t)
        feat_s, feat_t = feat_s + w_s[h] * cur_s, feat_t + w_t[h] * cur_t
    return torch.cat([feat_s, feat_t], dim=1)


def sgcn_conv(x, pos_ei, neg_ei, lin_b, lin_u, first_aggr, in_dim):
    """SGCNConv.forward in float64 (SGCNConv.py:94-126); lin_* = (weight [out, k in], bias or None)."""
    n = x.size(0)
    m_pos, m_neg = mean_operator(pos_ei, n), mean_operator(neg_ei, n)

    def lin(z, wb):
        out = z @ wb[0].t()
        return out if wb[1] is None else out + wb[1]

    if first_aggr:
        return torch.cat([lin(torch.cat([m_pos(x), x], 1), lin_b), lin(torch.cat([m_neg(x), x], 1), lin_u)], dim=1)
    lo, hi = x[:, :in_dim].contiguous(), x[:, in_dim:].contiguous()
    return torch.cat([lin(torch.cat([m_pos(lo), m_neg(hi), lo], 1), lin_b),
                      lin(torch.cat([m_pos(hi), m_neg(lo), hi], 1), lin_u)], dim=1)


def sssnet(ei_p, w_p, ei_n, w_n, features, sd, hop, fill, directed=False):
    """SSSNET_node_clustering.forward in eval mode (dropout = identity), float64; sd = its state_dict in float64.
    -> (normalize(z), log_softmax(output), softmax(output))."""
    streams = ("sp", "sn", "tp", "tn") if directed else ("p", "n")
    xs = [torch.relu(features @ sd[f"_w_{s}0"]) @ sd[f"_w_{s}1"] for s in streams]
    hop_weights = {k[len("_simpa."):]: v for k, v in sd.items() if k.startswith("_simpa.")}
    z = simpa(ei_p, w_p, ei_n, w_n, xs[0], xs[1], hop_weights, hop, fill, directed, *(xs[2:] if directed else ()))
    out = z @ sd["_W_prob"]
    if sd.get("_bias") is not None:
        out = out + sd["_bias"]
    return torch.nn.functional.normalize(z), torch.log_softmax(out, dim=1), torch.softmax(out, dim=1)
