"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Float64 SPARSE (scipy) evaluation of the MagNetConv / MSConv formulas, so that the HIP path can be checked at the
benchmark sizes (DSBM 100k / 2M in full, 1M / 20M on sampled rows) where oracle/dense_f64.py's N x N arrays do
not fit and oracle/ref_layers.py (fp32, the reference's op sequence) is itself only good to ~1e-6.

Same formulas as dense_f64.py -- MagNet paper eq. (1)-(3), reference utils/directed/get_magnetic_Laplacian.py:10-93
and utils/general/get_magnetic_signed_Laplacian.py:10-98 for the operator, nn/directed/MagNetConv.py:185-249
(quirk: out_real = A - B + b, out_imag = A + B + b with A = cheb(Re S^T, X_r), B = cheb(Im S^T, X_i)) for the
layer -- written with scipy.sparse matrix algebra: no sort / coalesce / scatter code shared with ref_layers.py or
the PyG shim (scipy's COO->CSR conversion sums duplicates itself).
"""
import numpy as np
import scipy.sparse as sp


def _hermitian_parts(edge_index, edge_weight, n):
    """One complex CSR C with C.real = A_s = (A + A^T) / 2 and C.imag = A - A^T on the SAME stored pattern
    (every listed non-loop edge u -> v adds w/2 + i w at (u, v) and w/2 - i w at (v, u); the COO -> CSR conversion
    sums duplicates), so magnitude and phase argument stay aligned entry by entry."""
    ei = np.asarray(edge_index)
    w = np.ones(ei.shape[1]) if edge_weight is None else np.asarray(edge_weight, np.float64)
    keep = ei[0] != ei[1]
    u, v, w = ei[0][keep], ei[1][keep], w[keep]
    data = np.concatenate([0.5 * w + 1j * w, 0.5 * w - 1j * w])
    return sp.coo_matrix((data, (np.concatenate([u, v]), np.concatenate([v, u]))), shape=(n, n)).tocsr()


def _degree(edge_index, edge_weight, n):
    """Row sums of A_s = (A + A^T) / 2 without forming it: (weighted out-degree + weighted in-degree) / 2."""
    ei = np.asarray(edge_index)
    w = np.ones(ei.shape[1]) if edge_weight is None else np.asarray(edge_weight, np.float64)
    keep = ei[0] != ei[1]
    return 0.5 * (np.bincount(ei[0][keep], w[keep], n) + np.bincount(ei[1][keep], w[keep], n))


def magnetic_operator(edge_index, edge_weight, n, q, normalization="sym", lambda_max=2.0, signed=False,
                      absolute_degree=True, only_nodes=None):
    """Complex CSR S = 2 L / lambda_max - I, L the (signed) magnetic Laplacian (float64 / complex128).

    only_nodes: build just the rows AND columns of S at these node ids (all other off-diagonal entries are left
    out): only the edges incident to them are assembled, the degrees still come from the whole edge list.  Enough
    for `magnet_conv_rows_k1` on the 1M-node benchmark graph, at a fraction of the full build's time."""
    ei = np.asarray(edge_index)
    if only_nodes is not None:
        if signed and not absolute_degree:
            raise ValueError("sampled build needs a degree that is a plain sum over the edge list")
        hit = np.zeros(n, dtype=bool)
        hit[np.asarray(only_nodes)] = True
        sel = hit[ei[0]] | hit[ei[1]]
        c = _hermitian_parts(ei[:, sel], None if edge_weight is None else np.asarray(edge_weight)[sel], n)
        w_deg = edge_weight if not signed or edge_weight is None else np.abs(np.asarray(edge_weight, np.float64))
        d = _degree(ei, w_deg, n)
    else:
        c = _hermitian_parts(ei, edge_weight, n)
        a_s = c.real
        if not signed:
            d = np.asarray(a_s.sum(1)).ravel()
        elif absolute_degree:
            w_abs = None if edge_weight is None else np.abs(np.asarray(edge_weight, np.float64))
            d = np.asarray(_hermitian_parts(ei, w_abs, n).real.sum(1)).ravel()
        else:
            d = np.asarray(abs(a_s).sum(1)).ravel()
    c = c.tocoo()
    mag, arg = c.data.real, c.data.imag
    if normalization == "sym":
        dis = np.zeros_like(d)
        dis[d != 0] = d[d != 0] ** -0.5
        mag = dis[c.row] * mag * dis[c.col]
        diag = np.ones(n)
    else:
        diag = d
    h = sp.coo_matrix((mag * np.exp(1j * (2.0 * np.pi * q) * arg), (c.row, c.col)), shape=(n, n)).tocsr()
    lap = sp.diags(diag.astype(np.complex128)) - h
    return (lap * (2.0 / lambda_max) - sp.identity(n, dtype=np.complex128)).tocsr()


def _cheb_terms(m, x, k1):
    ts = [x]
    if k1 > 1:
        ts.append(m @ x)
    for _ in range(2, k1):
        ts.append(2.0 * (m @ ts[-1]) - ts[-2])
    return ts


def magnet_conv(x_real, x_imag, s, weight, bias, g_real=None, g_imag=None):
    """Full evaluation (any K).  Returns (out_real, out_imag) and, when upstream gradients are given,
    also (dx_real, dx_imag, dweight, dbias) of  <out_real, g_real> + <out_imag, g_imag>."""
    xr, xi, w = (np.asarray(t, np.float64) for t in (x_real, x_imag, weight))
    k1 = w.shape[0]
    m_r, m_i = s.real.T.tocsr(), s.imag.T.tocsr()          # propagation aggregates at the column (target) index
    ta, tb = _cheb_terms(m_r, xr, k1), _cheb_terms(m_i, xi, k1)
    a = sum(ta[k] @ w[k] for k in range(k1))
    b = sum(tb[k] @ w[k] for k in range(k1))
    bb = 0.0 if bias is None else np.asarray(bias, np.float64)
    out = (a - b + bb, a + b + bb)
    if g_real is None:
        return out
    gr, gi = np.asarray(g_real, np.float64), np.asarray(g_imag, np.float64)
    p, mm = gr + gi, gi - gr                                # dL/dA, dL/dB
    dw = np.stack([ta[k].T @ p + tb[k].T @ mm for k in range(k1)])
    db = p.sum(0)                                           # d/db of (A - B + b, A + B + b) . (g_r, g_i) = sum(g_r + g_i)

    def adjoint(m, up):
        # gradient w.r.t. x of sum_k <T_k(m) x W_k, up>: run the Chebyshev recurrence with m^T on up W_k^T (Clenshaw)
        mt = m.T.tocsr()
        c = [up @ w[k].T for k in range(k1)]
        b1 = np.zeros_like(c[0])
        b2 = np.zeros_like(c[0])
        for k in range(k1 - 1, 0, -1):
            b1, b2 = c[k] + 2.0 * (mt @ b1) - b2, b1
        return c[0] + mt @ b1 - b2

    return out + (adjoint(m_r, p), adjoint(m_i, mm), dw, db)


def magnet_conv_rows_k1(x_real, x_imag, s, weight, bias, rows, g_real, g_imag):
    """K = 1 only, on a SAMPLE of node rows (the 1M-node benchmark graph): out_real / out_imag / dx_real / dx_imag
    restricted to `rows`; the sparse products touch only the sampled rows / columns of the operator, so `s` may be
    the partial operator `magnetic_operator(..., only_nodes=rows)`."""
    xr, xi, w = (np.asarray(t, np.float64) for t in (x_real, x_imag, weight))
    assert w.shape[0] == 2
    rows = np.asarray(rows)
    m_r, m_i = s.real.T.tocsr(), s.imag.T.tocsr()
    a = xr[rows] @ w[0] + (m_r[rows] @ xr) @ w[1]
    b = xi[rows] @ w[0] + (m_i[rows] @ xi) @ w[1]
    bb = 0.0 if bias is None else np.asarray(bias, np.float64)
    gr, gi = np.asarray(g_real, np.float64), np.asarray(g_imag, np.float64)
    p, mm = gr + gi, gi - gr
    dxr = p[rows] @ w[0].T + m_r.T.tocsr()[rows] @ (p @ w[1].T)
    dxi = mm[rows] @ w[0].T + m_i.T.tocsr()[rows] @ (mm @ w[1].T)
    return a - b + bb, a + b + bb, dxr, dxi
