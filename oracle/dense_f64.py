"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Independent float64 DENSE-matrix evaluation of the layer formulas documented in the
reference's docstrings / papers.  It shares no code and no
algorithm (no sort, no coalesce, no scatter) with oracle/ref_layers.py or with the PyG
shim: adjacency matrices are accumulated entry by entry into dense numpy arrays and
the layers are evaluated with dense matrix algebra.  It exists so that a mistake in
the shim or in the sparse restatement cannot silently define truth: golden fixtures
are only written after the reference-over-shim output agrees with this file.
"""
import numpy as np


def dense_adj(edge_index, edge_weight, n, drop_loops=False):
    """A[u, v] += w for every listed edge u -> v (duplicates add)."""
    a = np.zeros((n, n), dtype=np.float64)
    ei = np.asarray(edge_index)
    w = np.ones(ei.shape[1]) if edge_weight is None else np.asarray(edge_weight, dtype=np.float64)
    for (u, v), x in zip(ei.T, w):
        if drop_loops and u == v:
            continue
        a[u, v] += x
    return a


def pattern(edge_index, n, drop_loops=True):
    p = np.zeros((n, n), dtype=bool)
    for u, v in np.asarray(edge_index).T:
        if drop_loops and u == v:
            continue
        p[u, v] = True
    return p


def _inv_pow(d, p):
    out = np.zeros_like(d)
    nz = d != 0
    out[nz] = d[nz] ** p
    return out


def magnetic_operator(edge_index, edge_weight, n, q, normalization="sym", lambda_max=2.0,
                      signed=False, absolute_degree=True):
    """Complex dense S = 2 L / lambda_max - I with L the (signed) magnetic Laplacian.
    (MagNet paper eq. (1)-(3); MSGNN paper for the signed degree.)"""
    a = dense_adj(edge_index, edge_weight, n, drop_loops=True)
    a_s = (a + a.T) / 2
    if not signed:
        d = a_s.sum(1)
    elif absolute_degree:
        aa = dense_adj(edge_index, None if edge_weight is None else np.abs(edge_weight), n, True)
        d = ((aa + aa.T) / 2).sum(1)
    else:
        d = np.abs(a_s).sum(1)
    theta = 2 * np.pi * q * (a - a.T)
    # entries exist wherever u->v or v->u is listed (even if a_s cancels to 0)
    pat = pattern(edge_index, n)
    pat = pat | pat.T
    ph = np.where(pat, np.exp(1j * theta), 0)
    if normalization == "sym":
        dis = _inv_pow(d, -0.5)
        lap = np.eye(n) - (dis[:, None] * a_s * dis[None, :]) * ph
    else:
        lap = np.diag(d) - a_s * ph
    return 2.0 * lap / lambda_max - np.eye(n)


def cheb(s_t, x, weight):
    """sum_k T_k(s_t) x W_k."""
    t0 = x
    out = t0 @ weight[0]
    if weight.shape[0] > 1:
        t1 = s_t @ x
        out = out + t1 @ weight[1]
    for k in range(2, weight.shape[0]):
        t2 = 2 * (s_t @ t1) - t0
        out = out + t2 @ weight[k]
        t0, t1 = t1, t2
    return out


def magnet_conv(x_real, x_imag, s, weight, bias):
    """Quirk formula: A = cheb(Re S^T, X_r), B = cheb(Im S^T, X_i);
    out_real = A - B + b, out_imag = A + B + b."""
    x_real, x_imag, weight = (np.asarray(t, dtype=np.float64) for t in (x_real, x_imag, weight))
    a = cheb(s.real.T, x_real, weight)
    b = cheb(s.imag.T, x_imag, weight)
    bb = 0 if bias is None else np.asarray(bias, dtype=np.float64)
    return a - b + bb, a + b + bb


def digcn_conv(x, edge_index, edge_weight, weight, bias):
    n = x.shape[0]
    s = dense_adj(edge_index, edge_weight, n)
    out = s.T @ (np.asarray(x, np.float64) @ np.asarray(weight, np.float64))
    return out if bias is None else out + np.asarray(bias, np.float64)


def _with_remaining_loops(edge_index, edge_weight, n, fill):
    """A without its listed loops + diag(loop weight if a loop was listed (last wins) else fill)."""
    a = dense_adj(edge_index, edge_weight, n, drop_loops=True)
    diag = np.full(n, float(fill))
    ei = np.asarray(edge_index)
    w = np.ones(ei.shape[1]) if edge_weight is None else np.asarray(edge_weight, np.float64)
    for (u, v), x in zip(ei.T, w):
        if u == v:
            diag[u] = x
    return a + np.diag(diag)


def dgcn_conv(x, edge_index, edge_weight, improved=False, add_self_loops=True):
    n = x.shape[0]
    if add_self_loops:
        # current-PyG ordering: with edge_weight=None the loops get weight 1 regardless of improved
        fill = (2.0 if improved else 1.0) if edge_weight is not None else 1.0
        a = _with_remaining_loops(edge_index, edge_weight, n, fill)
    else:
        a = dense_adj(edge_index, edge_weight, n)
    dis = _inv_pow(a.sum(0), -0.5)  # degree over the target column
    s = dis[:, None] * a * dis[None, :]
    return s.T @ np.asarray(x, np.float64)


def conv_base_matrix(edge_index, edge_weight, n, fill, add_self_loops=True):
    a = _with_remaining_loops(edge_index, edge_weight, n, fill) if add_self_loops \
        else dense_adj(edge_index, edge_weight, n)
    return _inv_pow(a.sum(1), -1.0)[:, None] * a


def conv_base(x, edge_index, edge_weight, fill=0.5):
    return conv_base_matrix(edge_index, edge_weight, x.shape[0], fill) @ np.asarray(x, np.float64)


def _simpa_stream(ap, an, x_pos, x_neg, wp, wn, hop):
    """feat_p = sum_h wp[h] Ap^h x_pos ; feat_n = sum over (h, m) wn[.] Ap^m An Ap^h x_neg,
    h = 0..hop-1, m = 0..hop-1-h, ordered h-major (SSSNET paper, eq. for mixed-path)."""
    feat_p = wp[0] * x_pos
    cur = x_pos
    for h in range(1, hop + 1):
        cur = ap @ cur
        feat_p = feat_p + wp[h] * cur
    feat_n = np.zeros_like(feat_p)
    j = 0
    for h in range(hop):
        base = an @ np.linalg.matrix_power(ap, h) @ x_neg
        for m in range(hop - h):
            feat_n = feat_n + wn[j] * (np.linalg.matrix_power(ap, m) @ base)
            j += 1
    return feat_p, feat_n


def simpa(ei_p, w_p, ei_n, w_n, x_p, x_n, params, hop, fill, directed=False, x_pt=None, x_nt=None):
    n = x_p.shape[0]
    f = lambda t: np.asarray(t, np.float64)  # noqa: E731
    ap, an = conv_base_matrix(ei_p, w_p, n, fill), conv_base_matrix(ei_n, w_n, n, 0.0)
    if not directed:
        fp, fn = _simpa_stream(ap, an, f(x_p), f(x_n), f(params["_w_p"]).ravel(),
                               f(params["_w_n"]).ravel(), hop)
        return np.concatenate([fp, fn], 1)
    sp, sn = _simpa_stream(ap, an, f(x_p), f(x_n), f(params["_w_sp"]).ravel(),
                           f(params["_w_sn"]).ravel(), hop)
    flip = lambda e: np.asarray(e)[[1, 0]]  # noqa: E731
    apt, ant = conv_base_matrix(flip(ei_p), w_p, n, fill), conv_base_matrix(flip(ei_n), w_n, n, 0.0)
    tp, tn = _simpa_stream(apt, ant, f(x_pt), f(x_nt), f(params["_w_tp"]).ravel(),
                           f(params["_w_tn"]).ravel(), hop)
    return np.concatenate([sp, sn, tp, tn], 1)


def dimpa(x_s, x_t, edge_index, edge_weight, w_s, w_t, hop, fill=0.5):
    n = x_s.shape[0]
    a = conv_base_matrix(edge_index, edge_weight, n, fill)
    at = conv_base_matrix(np.asarray(edge_index)[[1, 0]], edge_weight, n, fill)
    w_s, w_t = np.asarray(w_s, np.float64).ravel(), np.asarray(w_t, np.float64).ravel()
    fs = sum(w_s[h] * (np.linalg.matrix_power(a, h) @ x_s) for h in range(hop + 1))
    ft = sum(w_t[h] * (np.linalg.matrix_power(at, h) @ x_t) for h in range(hop + 1))
    return np.concatenate([fs, ft], 1)


def mean_in(x, edge_index, n):
    """Row i = mean of x[j] over listed edges j -> i (multi-edges count twice; none -> 0)."""
    c = dense_adj(edge_index, None, n)  # c[j, i] = multiplicity of j -> i
    cnt = np.maximum(c.sum(0), 1)
    return (c.T @ np.asarray(x, np.float64)) / cnt[:, None]


def sgcn_conv(x, pos_ei, neg_ei, lin_b, lin_u, first_aggr, in_dim, norm_emb=False):
    x = np.asarray(x, np.float64)
    n = x.shape[0]
    lin = lambda z, wb: z @ np.asarray(wb[0], np.float64).T + (  # noqa: E731
        0 if wb[1] is None else np.asarray(wb[1], np.float64))
    if first_aggr:
        ob = lin(np.concatenate([mean_in(x, pos_ei, n), x], 1), lin_b)
        ou = lin(np.concatenate([mean_in(x, neg_ei, n), x], 1), lin_u)
    else:
        lo, hi = x[:, :in_dim], x[:, in_dim:]
        ob = lin(np.concatenate([mean_in(lo, pos_ei, n), mean_in(hi, neg_ei, n), lo], 1), lin_b)
        ou = lin(np.concatenate([mean_in(hi, pos_ei, n), mean_in(lo, neg_ei, n), hi], 1), lin_u)
    out = np.concatenate([ob, ou], 1)
    if norm_emb:
        out = out / np.maximum(np.linalg.norm(out, axis=1, keepdims=True), 1e-12)
    return out


def gat_conv(x, edge_index, lin_weight, att_src, att_dst, bias, negative_slope=0.2):
    """heads = 1 attention aggregate, node by node (no scatter, no segment ops)."""
    x = np.asarray(x, np.float64)
    n = x.shape[0]
    h = x @ np.asarray(lin_weight, np.float64).T
    a_s = h @ np.asarray(att_src, np.float64).reshape(-1)
    a_d = h @ np.asarray(att_dst, np.float64).reshape(-1)
    incoming = [[] for _ in range(n)]
    for u, v in np.asarray(edge_index).T:
        if u != v:
            incoming[v].append(u)
    out = np.zeros_like(h)
    for i in range(n):
        nb = incoming[i] + [i]                     # self loop re-added
        e = np.array([a_s[j] + a_d[i] for j in nb])
        e = np.where(e > 0, e, negative_slope * e)
        w = np.exp(e - e.max())
        w = w / w.sum()
        out[i] = (w[:, None] * h[nb]).sum(0)
    return out if bias is None else out + np.asarray(bias, np.float64)


def snea_conv(x, pos_ei, neg_ei, lin_b, lin_u, alpha_b, alpha_u, first_aggr, in_dim):
    """SNEAConv node by node (the message is the TARGET's row times the attention coefficient; self loops
    re-added only up to the largest node id left after loop removal)."""
    x = np.asarray(x, np.float64)
    n = x.shape[0]
    f = lambda t: np.asarray(t, np.float64)  # noqa: E731
    lin = lambda z, wb: z @ f(wb[0]).T + (0 if wb[1] is None else f(wb[1]))  # noqa: E731

    def incoming(ei, loops):
        pairs = [(int(u), int(v)) for u, v in np.asarray(ei).T if u != v]
        top = max([max(p) for p in pairs], default=-1) + 1 if loops else 0
        inc = [[] for _ in range(n)]
        for u, v in pairs:
            inc[v].append(u)
        for v in range(top):
            inc[v].append(v)
        return inc

    def aggregate(inc0, inc1, x1, x2, aw):
        w, b = f(aw[0]).reshape(-1), float(np.asarray(aw[1]).reshape(-1)[0])
        out = np.zeros_like(x1)
        for i in range(n):
            logits = [np.tanh(np.concatenate([x1[j], x1[i]]) @ w + b) for j in inc0[i]] + \
                     [np.tanh(np.concatenate([x2[j], x2[i]]) @ w + b) for j in inc1[i]]
            if not logits:
                continue
            e = np.exp(np.array(logits) - max(logits))
            a = e / e.sum()
            out[i] = x1[i] * a[:len(inc0[i])].sum() + x2[i] * a[len(inc0[i]):].sum()
        return out

    none = [[] for _ in range(n)]
    if first_aggr:
        hb, hu = lin(x, lin_b), lin(x, lin_u)
        ob = aggregate(incoming(pos_ei, True), none, hb, hb, alpha_b)
        ou = aggregate(incoming(neg_ei, True), none, hu, hu, alpha_u)
    else:
        hb, hu = x[:, :in_dim], x[:, in_dim:]
        inc0, inc1 = incoming(pos_ei, True), incoming(neg_ei, False)
        ob = aggregate(inc0, inc1, lin(hb, lin_b), lin(hu, lin_b), alpha_b)
        ou = aggregate(inc0, inc1, lin(hu, lin_u), lin(hb, lin_u), alpha_u)
    return np.concatenate([ob, ou], 1)
