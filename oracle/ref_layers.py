"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement ("port"; device-agnostic torch ops, so the tests at the stated sizes can also run the same op
sequence in fp32 on the test device) of the sparse message-passing hot path of
SherylHYX/pytorch_geometric_signed_directed, written as plain functions over
explicit parameter tensors.  It executes the SAME ATen op sequence the
reference executes through PyG on CPU -- per propagate
`x.index_select(0, src)` -> `w.view(-1,1) * x_j` -> `zeros.scatter_add_(0, dst, msg)`
-- including the reference's duplicated propagates, so it is both the parity
checker for the HIP path and the timed `cpu_baseline` (kind "port") of bench.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (pytorch_geometric_signed_directed_amd) never
does; it fails loudly if its HIP library is missing.

Parity pinning: the arithmetic of the path lives in `torch_geometric`
(un-vendored, un-pinned: reference setup.py:12; absent from this image), and the
reference's own tests hold no numeric vectors for this path.  This
restatement is therefore pinned against (a) golden vectors recorded by running the
reference's unmodified Python from /root/reference over a restated PyG shim
(oracle/pyg_shim, oracle/gen_golden.py -> tests/golden/*.npz), and (b) an
independent float64 dense-matrix evaluation of the documented formulas
(oracle/dense_f64.py) plus the hand-checked known-answer vector of SURVEY.md
Appendix B.  tests/test_oracle_golden.py enforces both.

Reference files restated here (all under torch_geometric_signed_directed/):
  nn/directed/MagNetConv.py, nn/general/MSConv.py,
  utils/directed/get_magnetic_Laplacian.py, utils/general/get_magnetic_signed_Laplacian.py,
  nn/directed/DiGCNConv.py, nn/directed/DGCNConv.py, nn/general/conv_base.py,
  nn/signed/SIMPA.py, nn/directed/DIMPA.py, nn/signed/SGCNConv.py,
  nn/directed/complex_relu.py
and the PyG primitives they call (semantics: SURVEY.md Appendix A).
"""
import math
from typing import Optional, Tuple

import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# PyG primitives (third-party, restated from published semantics)
# --------------------------------------------------------------------------
def scatter_rows(msg: Tensor, dst: Tensor, n_out: int, reduce: str = "add") -> Tensor:
    """torch_geometric.utils.scatter along dim 0: add, or mean with count clamped >= 1."""
    out = msg.new_zeros((n_out,) + tuple(msg.shape[1:]))
    idx = dst.view((-1,) + (1,) * (msg.dim() - 1)).expand_as(msg)
    out.scatter_add_(0, idx, msg)
    if reduce in ("add", "sum"):
        return out
    if reduce == "mean":
        cnt = msg.new_zeros(n_out).scatter_add_(0, dst, msg.new_ones(dst.numel()))
        return out / cnt.clamp(min=1).view((-1,) + (1,) * (msg.dim() - 1))
    raise ValueError(reduce)


def propagate(x: Tensor, edge_index: Tensor, w: Optional[Tensor], n_out: int,
              flow: str = "source_to_target", reduce: str = "add") -> Tensor:
    """MessagePassing.propagate for message = w * x_j (or x_j): gather, scale, scatter.

    source_to_target: out[edge_index[1,e]] += w[e] * x[edge_index[0,e]]
    target_to_source: out[edge_index[0,e]] += w[e] * x[edge_index[1,e]]
    (reference message(): MagNetConv.py:251, DiGCNConv.py:88, DGCNConv.py:99,
    conv_base.py:116, SGCNConv.py:128)
    """
    gather_row, scatter_row = (0, 1) if flow == "source_to_target" else (1, 0)
    msg = x.index_select(0, edge_index[gather_row])
    if w is not None:
        msg = w.view(-1, 1) * msg
    return scatter_rows(msg, edge_index[scatter_row], n_out, reduce)


def drop_self_loops(edge_index: Tensor, attr: Optional[Tensor]):
    keep = edge_index[0] != edge_index[1]
    return edge_index[:, keep], (None if attr is None else attr[keep])


def append_self_loops(edge_index: Tensor, attr: Optional[Tensor], fill: float, n: int):
    loops = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device).expand(2, n)
    ei = torch.cat([edge_index, loops], dim=1)
    if attr is None:
        return ei, None
    tail = attr.new_full((n,) + tuple(attr.shape[1:]), fill)
    return ei, torch.cat([attr, tail], dim=0)


def append_remaining_self_loops(edge_index: Tensor, attr: Optional[Tensor], fill: float, n: int):
    """Existing loops are pulled out and re-appended as the N trailing entries; a node that had
    a loop keeps that loop's weight (last one wins), every other node gets `fill`."""
    off = edge_index[0] != edge_index[1]
    loops = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device).expand(2, n)
    ei = torch.cat([edge_index[:, off], loops], dim=1)
    if attr is None:
        return ei, None
    tail = attr.new_full((n,), fill)
    on = ~off
    tail[edge_index[0][on]] = attr[on]
    return ei, torch.cat([attr[off], tail], dim=0)


def coalesce_add(edge_index: Tensor, attr: Tensor, n: int):
    """Sort by (row, col), merge duplicates by summation (torch_geometric.utils.coalesce)."""
    key = edge_index[0] * n + edge_index[1]
    key, order = key.sort(stable=True)
    ei = edge_index[:, order]
    attr = attr[order]
    head = torch.ones_like(key, dtype=torch.bool)
    head[1:] = key[1:] != key[:-1]
    seg = head.long().cumsum(0) - 1
    ei = ei[:, head]
    return ei, scatter_rows(attr, seg, ei.size(1), "add")


# --------------------------------------------------------------------------
# a3 / a4: magnetic Laplacian operator build
# --------------------------------------------------------------------------
def magnetic_laplacian(edge_index: Tensor, edge_weight: Optional[Tensor], n: int, q,
                       normalization: Optional[str] = "sym", signed: bool = False,
                       absolute_degree: bool = True, dtype=torch.float32):
    """get_magnetic_Laplacian.py:10-93 / get_magnetic_signed_Laplacian.py:10-98.

    Returns (edge_index [2, E_s + N], real [E_s + N], imag [E_s + N]); entries sorted by
    (row, col) followed by N self loops.
    """
    edge_index, edge_weight = drop_self_loops(edge_index, edge_weight)
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype, device=edge_index.device)
    r, c = edge_index[0], edge_index[1]
    both = torch.stack([torch.cat([r, c]), torch.cat([c, r])])
    cols = [torch.cat([edge_weight, edge_weight]), torch.cat([edge_weight, -edge_weight])]
    if signed:
        cols.append(torch.cat([edge_weight.abs(), edge_weight.abs()]))
    sym_index, attr = coalesce_add(both, torch.stack(cols, dim=1), n)
    row, col = sym_index[0], sym_index[1]
    a_sym = attr[:, 0] / 2
    if not signed:
        deg = scatter_rows(a_sym, row, n)
    elif absolute_degree:
        deg = scatter_rows(attr[:, 2] / 2, row, n)
    else:
        deg = scatter_rows(a_sym.abs(), row, n)
    phase = torch.exp(1j * 2 * math.pi * q * attr[:, 1])
    if normalization is None:
        out_index, _ = append_self_loops(sym_index, None, 1.0, n)
        lap = torch.cat([-a_sym * phase, deg], dim=0)
    else:
        assert normalization == "sym"
        dis = deg.pow(-0.5)
        dis = dis.masked_fill(dis == float("inf"), 0)
        h = dis[row] * a_sym * dis[col] * phase
        out_index, lap = append_self_loops(sym_index, -h, 1.0, n)
    return out_index, lap.real, lap.imag


def laplacian_lambda_max(edge_index, edge_weight, n, q, signed=False, absolute_degree=True):
    """normalization=None branch: largest-magnitude eigenvalue via scipy eigsh
    (get_magnetic_Laplacian.py:88-92)."""
    import numpy as np
    import scipy.sparse as sp
    from scipy.sparse.linalg import eigsh
    ei, re, im = magnetic_laplacian(edge_index, edge_weight, n, q, None, signed, absolute_degree)
    val = (re.to(torch.complex64) + 1j * im.to(torch.complex64)).numpy()
    L = sp.coo_matrix((val, (ei[0].numpy(), ei[1].numpy())), (n, n))
    lam = eigsh(L, k=1, which="LM", return_eigenvectors=False)
    return float(np.asarray(lam).real.item())


def magnet_operator(edge_index, edge_weight, n, q, normalization, lambda_max,
                    signed=False, absolute_degree=True, dtype=torch.float32):
    """MagNetConv.__norm__ (MagNetConv.py:78-120) / MSConv.__norm__ (MSConv.py:78-119):
    scaled operator 2L/lambda_max - I as two COO operators.

    real: E_s off-diagonals, N loops (2/lambda), N loops (-1);  imag: E_s off-diagonals, N zeros.
    """
    edge_index, edge_weight = drop_self_loops(edge_index, edge_weight)
    ei, re, im = magnetic_laplacian(edge_index, edge_weight, n, q, normalization,
                                    signed, absolute_degree, dtype)
    lam = torch.as_tensor(lambda_max, dtype=dtype)
    re = (2.0 * re) / lam
    re = re.masked_fill(re == float("inf"), 0)
    ei_real, re = append_self_loops(ei, re, -1.0, n)
    im = (2.0 * im) / lam
    im = im.masked_fill(im == float("inf"), 0)
    return ei_real, ei.clone(), re, im


# --------------------------------------------------------------------------
# a1 / a2: MagNetConv / MSConv forward
# --------------------------------------------------------------------------
def cheb_chain(x: Tensor, edge_index: Tensor, norm: Tensor, weight: Tensor) -> Tensor:
    """sum_k T_k(S^T) x W_k with T_0 = x, T_1 = S^T x, T_k = 2 S^T T_{k-1} - T_{k-2}
    (one of the four chains of MagNetConv.py:185-240)."""
    n = x.size(0)
    t_prev = x
    out = torch.matmul(t_prev, weight[0])
    if weight.size(0) > 1:
        t_cur = propagate(x, edge_index, norm, n)
        out = out + torch.matmul(t_cur, weight[1])
    for k in range(2, weight.size(0)):
        t_next = propagate(t_cur, edge_index, norm, n)
        t_next = 2.0 * t_next - t_prev
        out = out + torch.matmul(t_next, weight[k])
        t_prev, t_cur = t_cur, t_next
    return out


def magnet_conv(x_real, x_imag, operator, weight, bias, duplicate=True):
    """MagNetConv.forward / MSConv.forward given a built operator
    (MagNetConv.py:185-249).  `duplicate=True` evaluates the four chains the reference
    evaluates (two are exact duplicates) -- this is what the timed CPU baseline runs."""
    ei_r, ei_i, w_r, w_i = operator
    rr = cheb_chain(x_real, ei_r, w_r, weight)
    ii = cheb_chain(x_imag, ei_i, w_i, weight)
    if duplicate:
        ir = cheb_chain(x_real, ei_r, w_r, weight)
        ri = cheb_chain(x_imag, ei_i, w_i, weight)
    else:
        ir, ri = rr, ii
    out_real = rr - ii
    out_imag = ir + ri
    if bias is not None:
        out_real = out_real + bias
        out_imag = out_imag + bias
    return out_real, out_imag


def complex_relu(real, imag):
    """complex_relu.py:21-22."""
    mask = 1.0 * (real >= 0)
    return mask * real, mask * imag


# --------------------------------------------------------------------------
# a5 / a7 / a8: DiGCNConv, DGCNConv, Conv_Base
# --------------------------------------------------------------------------
def digcn_conv(x, edge_index, edge_weight, weight, bias):
    """DiGCNConv.forward (DiGCNConv.py:54-94): S^T (x W) + b."""
    h = torch.matmul(x, weight)
    out = propagate(h, edge_index, edge_weight, x.size(0))
    return out if bias is None else out + bias


def gcn_norm(edge_index, edge_weight, n, improved=False, add_self_loops=True,
             dtype=torch.float32):
    """torch_geometric.nn.conv.gcn_conv.gcn_norm (current ordering: loops first, then
    default ones), degree over the target column."""
    if add_self_loops:
        edge_index, edge_weight = append_remaining_self_loops(
            edge_index, edge_weight, 2.0 if improved else 1.0, n)
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype, device=edge_index.device)
    row, col = edge_index[0], edge_index[1]
    deg = scatter_rows(edge_weight, col, n)
    dis = deg.pow(-0.5)
    dis = dis.masked_fill(dis == float("inf"), 0)
    return edge_index, dis[row] * edge_weight * dis[col]


def dgcn_conv(x, edge_index, edge_weight, improved=False, add_self_loops=True, normalize=True):
    """DGCNConv.forward edge_index path (DGCNConv.py:60-97)."""
    if normalize:
        edge_index, edge_weight = gcn_norm(edge_index, edge_weight, x.size(0), improved,
                                           add_self_loops, x.dtype)
    return propagate(x, edge_index, edge_weight, x.size(0))


def conv_norm_rw(edge_index, edge_weight, n, fill_value=0.5, add_self_loops=True,
                 dtype=torch.float32):
    """conv_base.py:12-31: row-normalised D^-1 (A + fill I)."""
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype, device=edge_index.device)
    if add_self_loops:
        edge_index, edge_weight = append_remaining_self_loops(edge_index, edge_weight,
                                                              fill_value, n)
    row = edge_index[0]
    deg = scatter_rows(edge_weight, row, n)
    inv = deg.pow(-1)
    inv = inv.masked_fill(inv == float("inf"), 0)
    return edge_index, inv[row] * edge_weight


def conv_base(x, edge_index, edge_weight, fill_value=0.5, add_self_loops=True, normalize=True):
    """Conv_Base.forward (conv_base.py:98-114): flow=target_to_source, never cached."""
    if normalize:
        edge_index, edge_weight = conv_norm_rw(edge_index, edge_weight, x.size(0), fill_value,
                                               add_self_loops, x.dtype)
    return propagate(x, edge_index, edge_weight, x.size(0), flow="target_to_source")


# --------------------------------------------------------------------------
# a9: SIMPA / DIMPA hop schedules
# --------------------------------------------------------------------------
def _simpa_stream(ei_p, w_p, ei_n, w_n, x_pos, x_neg, wp, wn, hop, fill):
    """One (positive, negative) feature pair of SIMPA (SIMPA.py:77-93 / :101-139)."""
    hop_p = hop + 1
    feat_p = wp[0] * x_pos
    feat_n = torch.zeros_like(feat_p)
    cur_p, aux_n = x_pos.clone(), x_neg.clone()
    j = 0
    for h in range(hop_p):
        if h > 0:
            cur_p = conv_base(cur_p, ei_p, w_p, fill)
            aux_n = conv_base(aux_n, ei_p, w_p, fill)
            feat_p = feat_p + wp[h] * cur_p
        if h != hop_p - 1:
            cur_n = conv_base(aux_n, ei_n, w_n, 0.0)
            feat_n = feat_n + wn[j] * cur_n
            j += 1
            for _ in range(hop_p - 2 - h):
                cur_n = conv_base(cur_n, ei_p, w_p, fill)
                feat_n = feat_n + wn[j] * cur_n
                j += 1
    return feat_p, feat_n


def simpa(ei_p, w_p, ei_n, w_n, x_p, x_n, params, hop, fill_value, directed=False,
          x_pt=None, x_nt=None):
    """SIMPA.forward (SIMPA.py:52-144).  params: dict of the module's weights."""
    if not directed:
        fp, fn = _simpa_stream(ei_p, w_p, ei_n, w_n, x_p, x_n, params["_w_p"], params["_w_n"],
                               hop, fill_value)
        return torch.cat([fp, fn], dim=1)
    sp, sn = _simpa_stream(ei_p, w_p, ei_n, w_n, x_p, x_n, params["_w_sp"], params["_w_sn"],
                           hop, fill_value)
    tp, tn = _simpa_stream(ei_p[[1, 0]], w_p, ei_n[[1, 0]], w_n, x_pt, x_nt,
                           params["_w_tp"], params["_w_tn"], hop, fill_value)
    return torch.cat([sp, sn, tp, tn], dim=1)


def dimpa(x_s, x_t, edge_index, edge_weight, w_s, w_t, hop, fill_value=0.5):
    """DIMPA.forward (DIMPA.py:32-59)."""
    feat_s, feat_t = w_s[0] * x_s, w_t[0] * x_t
    cur_s, cur_t = x_s.clone(), x_t.clone()
    ei_t = edge_index[[1, 0]]
    for h in range(1, hop + 1):
        cur_s = conv_base(cur_s, edge_index, edge_weight, fill_value)
        cur_t = conv_base(cur_t, ei_t, edge_weight, fill_value)
        feat_s = feat_s + w_s[h] * cur_s
        feat_t = feat_t + w_t[h] * cur_t
    return torch.cat([feat_s, feat_t], dim=1)


# --------------------------------------------------------------------------
# a10: SGCNConv
# --------------------------------------------------------------------------
def sgcn_conv(x, pos_edge_index, neg_edge_index, lin_b: Tuple[Tensor, Optional[Tensor]],
              lin_u: Tuple[Tensor, Optional[Tensor]], first_aggr: bool, in_dim: int,
              norm_emb: bool = False):
    """SGCNConv.forward (SGCNConv.py:94-126): mean-aggregate over pos / neg incoming edges."""
    import torch.nn.functional as F
    n = x.size(0)

    def mean_in(feat, ei):
        return propagate(feat, ei, None, n, reduce="mean")

    if first_aggr:
        out_b = F.linear(torch.cat([mean_in(x, pos_edge_index), x], dim=-1), *lin_b)
        out_u = F.linear(torch.cat([mean_in(x, neg_edge_index), x], dim=-1), *lin_u)
    else:
        lo, hi = x[..., :in_dim], x[..., in_dim:]
        out_b = F.linear(torch.cat([mean_in(lo, pos_edge_index), mean_in(hi, neg_edge_index), lo],
                                   dim=-1), *lin_b)
        out_u = F.linear(torch.cat([mean_in(hi, pos_edge_index), mean_in(lo, neg_edge_index), hi],
                                   dim=-1), *lin_u)
    out = torch.cat([out_b, out_u], dim=-1)
    return F.normalize(out, p=2, dim=-1) if norm_emb else out


# --------------------------------------------------------------------------
# a13: attention aggregate of SDGNN / SiGAT (third-party torch_geometric.nn.GATConv, heads >= 1)
# --------------------------------------------------------------------------
def segment_softmax(e: Tensor, index: Tensor, n: int) -> Tensor:
    """torch_geometric.utils.softmax: per-target max-shifted exp / (segment sum + 1e-16)."""
    shape = (n,) + tuple(e.shape[1:])
    idx = index.view((-1,) + (1,) * (e.dim() - 1)).expand_as(e)
    mx = torch.full(shape, float("-inf"), dtype=e.dtype, device=e.device).scatter_reduce(0, idx, e.detach(), "amax", include_self=True)
    out = (e - mx.index_select(0, index)).exp()
    den = torch.zeros(shape, dtype=e.dtype, device=e.device).scatter_add_(0, idx, out) + 1e-16
    return out / den.index_select(0, index)


def gat_conv(x, edge_index, lin_weight, att_src, att_dst, bias, heads=1, concat=True, negative_slope=0.2,
             add_self_loops=True):
    """GATConv as SDRLayer calls it (reference nn/signed/SDGNN.py:35-41,57-64; SiGAT.py:59-64)."""
    import torch.nn.functional as F
    n = x.size(0)
    c = lin_weight.size(0) // heads
    h = F.linear(x, lin_weight).view(-1, heads, c)
    a_src = (h * att_src).sum(-1)
    a_dst = (h * att_dst).sum(-1)
    if add_self_loops:
        edge_index, _ = drop_self_loops(edge_index, None)
        edge_index, _ = append_self_loops(edge_index, None, 1.0, n)
    j, i = edge_index[0], edge_index[1]
    alpha = segment_softmax(F.leaky_relu(a_src[j] + a_dst[i], negative_slope), i, n)
    out = scatter_rows(alpha.unsqueeze(-1) * h[j], i, n)
    out = out.reshape(n, heads * c) if concat else out.mean(dim=1)
    return out if bias is None else out + bias
