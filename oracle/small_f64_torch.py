"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Float64 evaluations WITH GRADIENTS (torch autograd) of the toy-sized pieces whose recorded fixtures hold fp32 results of
the reference only: SNEAConv node by node, the SSSNET cut objectives and DIGRAC's imbalance objective as dense formulas.
They are the float64 arbiter of the GPU checks that compare a HIP result with such a fixture (tests/tolerance.py
close_arbitrated): both fp32 results -- the recorded reference's and the HIP path's -- are measured against these.
Pinned on the host (tests/test_oracle_small_f64.py): forward values against oracle/dense_f64.py's numpy formulas where
one exists, values and gradients against the recorded reference fixtures (<= 2e-5).

Reference formulas: nn/signed/SNEAConv.py:70-146; utils/signed/prob_balanced_normalized_loss.py:16-48,
prob_balanced_ratio_loss.py:16-42, unhappy_ratio.py:14-45; utils/directed/prob_imbalance_loss.py:27-117.
"""
import numpy as np
import torch

F64 = torch.float64


def _t(a):
    return a.double() if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a, np.float64))


def snea_conv(x, pos_ei, neg_ei, lin_b, lin_u, alpha_b, alpha_u, first_aggr, in_dim):
    """SNEAConv node by node (the message is the TARGET's row times the attention coefficient; self loops re-added only
    up to the largest node id left after loop removal).  Parameters: float64 tensors (requires_grad allowed)."""
    n = x.size(0)
    lin = lambda z, wb: z @ wb[0].t() + (0 if wb[1] is None else wb[1])  # noqa: E731

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
        w, b = aw[0].reshape(-1), aw[1].reshape(-1)[0]
        rows = []
        for i in range(n):
            logits = [torch.tanh(torch.cat([x1[j], x1[i]]) @ w + b) for j in inc0[i]] + \
                     [torch.tanh(torch.cat([x2[j], x2[i]]) @ w + b) for j in inc1[i]]
            if not logits:
                rows.append(torch.zeros_like(x1[i]))
                continue
            a = torch.softmax(torch.stack(logits), 0)
            rows.append(x1[i] * a[:len(inc0[i])].sum() + x2[i] * a[len(inc0[i]):].sum())
        return torch.stack(rows)

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
    return torch.cat([ob, ou], 1)


def cut_losses(a_p, a_n, prob):
    """(balanced normalised cut, balanced ratio cut, unhappy ratio) of SSSNET for dense float64 A_p, A_n [n, n] and a
    probability matrix [n, K]: sum_k p_k^T (D_p - A_p + A_n) p_k / (p_k^T (D_p + D_n) p_k + 1e-6), the same numerator over
    p_k^T p_k + 1, and the numerators' sum over the number of stored entries of A_p - A_n."""
    a_p, a_n, p = _t(a_p), _t(a_n), prob
    d_p, d_n = torch.diag(a_p.sum(1)), torch.diag(a_n.sum(1))
    mat, d_bar = d_p - (a_p - a_n), d_p + d_n
    num = torch.stack([p[:, k] @ mat @ p[:, k] for k in range(p.size(1))])
    den_n = torch.stack([p[:, k] @ d_bar @ p[:, k] for k in range(p.size(1))]) + 1e-6
    den_r = (p * p).sum(0) + 1.0
    edges = int(((a_p - a_n) != 0).sum())
    return (num / den_n).sum(), (num / den_r).sum(), num.sum() / edges


def imbalance_loss(prob, adj, k, sel, normalization="vol_sum", threshold="sort"):
    """1 - (mean of the selected pairwise imbalance scores) of DIGRAC for a dense float64 adjacency [n, n]
    (prob_imbalance_loss.py:27-117; sel = number of pairs kept by the 'sort' threshold)."""
    a, p = _t(adj), prob
    eps = 1e-8
    vol = torch.stack([((a + a.t()) @ p[:, c:c + 1]).sum() for c in range(k)])
    second = torch.topk(vol, 2).values[1] + eps
    kept, below = [], []
    for c in range(k - 1):
        for l in range(c + 1, k):
            w_cl, w_lc = p[:, c] @ a @ p[:, l], p[:, l] @ a @ p[:, c]
            diff, tot = w_cl - w_lc, w_cl + w_lc
            if float(diff.detach()) == 0:
                continue
            if normalization == "vol_sum":
                cur = diff.abs() / (vol[c] + vol[l] + eps) * 2
            elif normalization == "vol_min":
                cur = diff.abs() / tot * torch.min(vol[c], vol[l]) / second
            elif normalization == "vol_max":
                cur = diff.abs() / (torch.max(vol[c], vol[l]) + eps)
            else:
                cur = diff.abs() / tot
            (kept if threshold != "std" or float(diff.detach()) ** 2 - 9 * float(tot.detach()) > 0 else below).append(cur)
    one = torch.ones(1, dtype=F64)
    if threshold == "sort":
        order = np.argsort(-np.array([float(c.detach()) for c in kept]))
        return one - sum(kept[i] for i in order[:int(sel)]) / sel
    if kept:
        return one - torch.stack(kept).mean().detach()       # the reference re-wraps the values: no gradient
    if threshold == "std":
        return one - torch.stack(below).mean().detach()
    return one
