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
