"""ORACLE -- TEST INFRASTRUCTURE.  Generates tests/golden/*.npz.  Runs ONLY in the build
container (needs /root/reference); the GPU box never runs it.

    python oracle/gen_golden.py            # regenerate every fixture

For every case it (1) runs the reference's own, unmodified Python from /root/reference
(imported over oracle/pyg_shim, because `torch_geometric` is absent/un-pinned), (2) checks the
result against the independent float64 dense formulas of oracle/dense_f64.py (<= 2e-6 * scale)
and refuses to write the fixture otherwise, (3) records inputs, parameters, the operator the
layer built, outputs and input/parameter gradients.  Fixtures are data only.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "pyg_shim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from torch_geometric_signed_directed.nn.directed.MagNetConv import MagNetConv  # noqa: E402
from torch_geometric_signed_directed.nn.general.MSConv import MSConv  # noqa: E402
from torch_geometric_signed_directed.nn.directed.DiGCNConv import DiGCNConv  # noqa: E402
from torch_geometric_signed_directed.nn.directed.DGCNConv import DGCNConv  # noqa: E402
from torch_geometric_signed_directed.nn.general.conv_base import Conv_Base  # noqa: E402
from torch_geometric_signed_directed.nn.signed.SIMPA import SIMPA  # noqa: E402
from torch_geometric_signed_directed.nn.directed.DIMPA import DIMPA  # noqa: E402
from torch_geometric_signed_directed.nn.signed.SGCNConv import SGCNConv  # noqa: E402
from torch_geometric_signed_directed.nn.directed.complex_relu import complex_relu_layer  # noqa: E402
from torch_geometric_signed_directed.utils.directed.get_magnetic_Laplacian import \
    get_magnetic_Laplacian  # noqa: E402
from torch_geometric_signed_directed.utils.general.get_magnetic_signed_Laplacian import \
    get_magnetic_signed_Laplacian  # noqa: E402

from oracle import dense_f64 as D  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
TOL = 2e-6


def toy_graph(seed, n=40, e=170, isolated=3, loops=4, dups=6, recip=10, signed=False,
              weighted=True):
    """Small digraph with every structural edge case: self loops, multi-edges, reciprocal
    pairs, isolated nodes (the last `isolated` ids never appear)."""
    rng = np.random.default_rng(seed)
    m = n - isolated
    src = rng.integers(0, m, e)
    dst = rng.integers(0, m, e)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    src = np.concatenate([src, dst[:recip], src[:dups], rng.integers(0, m, loops)])
    dst = np.concatenate([dst, src[:recip], dst[:dups], src[-loops:]])
    perm = rng.permutation(src.size)
    ei = np.stack([src[perm], dst[perm]]).astype(np.int64)
    w = rng.uniform(0.5, 2.0, ei.shape[1]).astype(np.float32)
    if signed:
        w *= rng.choice([-1.0, 1.0], ei.shape[1]).astype(np.float32)
    return ei, (w if weighted else None)


def close(name, got, want, scale=None):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    s = max(1.0, float(np.abs(want).max())) if scale is None else scale
    err = float(np.abs(got - want).max()) / s
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert err <= TOL, f"{name}: reference-over-shim vs dense float64 differ by {err:.3e}"
    return err


def t(x):
    return None if x is None else torch.from_numpy(np.asarray(x))


def npy(x):
    return x.detach().cpu().numpy()


def save(name, **arrs):
    arrs = {k: v for k, v in arrs.items() if v is not None}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print(f"  wrote {name}.npz ({len(arrs)} arrays)")


# ------------------------------------------------------------------ MagNetConv / MSConv
def magnet_case(name, seed, K, normalization, weighted, signed=False, absolute_degree=True,
                fin=6, fout=5, q=0.25, bias=True):
    ei, w = toy_graph(seed, signed=signed, weighted=weighted)
    n = 40
    g = torch.Generator().manual_seed(seed)
    xr = torch.randn(n, fin, generator=g)
    xi = torch.randn(n, fin, generator=g)
    gr = torch.randn(n, fout, generator=g)
    gi = torch.randn(n, fout, generator=g)
    torch.manual_seed(seed)
    if signed:
        layer = MSConv(fin, fout, K, q, False, normalization=normalization, bias=bias,
                       absolute_degree=absolute_degree)
    else:
        layer = MagNetConv(fin, fout, K, q, False, normalization=normalization, bias=bias)
    if bias:
        with torch.no_grad():
            layer.bias.uniform_(-0.5, 0.5)
    lam = None
    if normalization is None:
        fn = get_magnetic_signed_Laplacian if signed else get_magnetic_Laplacian
        kw = dict(absolute_degree=absolute_degree) if signed else {}
        lam = fn(t(ei), t(w), None, q=q, return_lambda_max=True, **kw)[3]
    xr.requires_grad_(True)
    xi.requires_grad_(True)
    out_r, out_i = layer(xr, xi, t(ei), t(w), lambda_max=lam)
    ((out_r * gr).sum() + (out_i * gi).sum()).backward()
    op = layer.cached_result
    # independent check
    S = D.magnetic_operator(ei, w, n, q, normalization, 2.0 if lam is None else lam, signed,
                            absolute_degree)
    dr, di = D.magnet_conv(npy(xr), npy(xi), S, npy(layer.weight),
                           npy(layer.bias) if bias else None)
    close(name + ".out_real", npy(out_r), dr)
    close(name + ".out_imag", npy(out_i), di)
    # the recorded operator must equal the dense S entry-wise
    for ei_k, val, part in ((op[0], op[2], S.real), (op[1], op[3], S.imag)):
        acc = np.zeros((n, n))
        np.add.at(acc, (npy(ei_k[0]), npy(ei_k[1])), npy(val).astype(np.float64))
        close(name + ".operator", acc, part)
    save(name, edge_index=ei, edge_weight=w, x_real=npy(xr), x_imag=npy(xi), grad_real=npy(gr),
         grad_imag=npy(gi), weight=npy(layer.weight), bias=npy(layer.bias) if bias else None,
         q=np.float64(q), K=np.int64(K), lambda_max=None if lam is None else np.float64(lam),
         normalization=np.array("none" if normalization is None else normalization),
         signed=np.bool_(signed), absolute_degree=np.bool_(absolute_degree),
         op_index_real=npy(op[0]), op_index_imag=npy(op[1]), op_real=npy(op[2]), op_imag=npy(op[3]),
         out_real=npy(out_r), out_imag=npy(out_i), dx_real=npy(xr.grad), dx_imag=npy(xi.grad),
         dweight=npy(layer.weight.grad), dbias=npy(layer.bias.grad) if bias else None)


# ------------------------------------------------------------------ DiGCNConv / DGCNConv / Conv_Base
def digcn_case(name, seed, fin=7, fout=4, bias=True):
    ei, w = toy_graph(seed)
    n = 40
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, fin, generator=g, requires_grad=True)
    go = torch.randn(n, fout, generator=g)
    torch.manual_seed(seed)
    layer = DiGCNConv(fin, fout, bias=bias)
    if bias:
        with torch.no_grad():
            layer.bias.uniform_(-0.5, 0.5)
    out = layer(x, t(ei), t(w))
    (out * go).sum().backward()
    close(name, npy(out), D.digcn_conv(npy(x), ei, w, npy(layer.weight),
                                       npy(layer.bias) if bias else None))
    save(name, edge_index=ei, edge_weight=w, x=npy(x), grad_out=npy(go), weight=npy(layer.weight),
         bias=npy(layer.bias) if bias else None, out=npy(out), dx=npy(x.grad),
         dweight=npy(layer.weight.grad), dbias=npy(layer.bias.grad) if bias else None)


def dgcn_case(name, seed, weighted, improved, add_self_loops, f=6):
    ei, w = toy_graph(seed, weighted=weighted)
    n = 40
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, f, generator=g, requires_grad=True)
    go = torch.randn(n, f, generator=g)
    layer = DGCNConv(improved=improved, add_self_loops=add_self_loops)
    out = layer(x, t(ei), t(w))
    (out * go).sum().backward()
    close(name, npy(out), D.dgcn_conv(npy(x), ei, w, improved, add_self_loops))
    save(name, edge_index=ei, edge_weight=w, x=npy(x), grad_out=npy(go), out=npy(out),
         dx=npy(x.grad), improved=np.bool_(improved), add_self_loops=np.bool_(add_self_loops))


def conv_base_case(name, seed, weighted, fill, f=6):
    ei, w = toy_graph(seed, weighted=weighted)
    n = 40
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, f, generator=g, requires_grad=True)
    go = torch.randn(n, f, generator=g)
    layer = Conv_Base(fill)
    out = layer(x, t(ei), t(w))
    (out * go).sum().backward()
    close(name, npy(out), D.conv_base(npy(x), ei, w, fill))
    save(name, edge_index=ei, edge_weight=w, x=npy(x), grad_out=npy(go), out=npy(out),
         dx=npy(x.grad), fill_value=np.float64(fill))


# ------------------------------------------------------------------ SIMPA / DIMPA
def simpa_case(name, seed, hop, directed, fill=0.5, f=5):
    ei_p, w_p = toy_graph(seed)
    ei_n, w_n = toy_graph(seed + 100, e=90)
    n = 40
    g = torch.Generator().manual_seed(seed)
    xs = [torch.randn(n, f, generator=g, requires_grad=True) for _ in range(4)]
    layer = SIMPA(hop, fill, directed)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(torch.rand(p.shape, generator=g) + 0.5)
    params = {k: npy(v) for k, v in layer.state_dict().items()}
    args = (t(ei_p), t(w_p), t(ei_n), t(w_n), xs[0], xs[1]) + ((xs[2], xs[3]) if directed else ())
    out = layer(*args)
    go = torch.randn(out.shape, generator=g)
    (out * go).sum().backward()
    close(name, npy(out), D.simpa(ei_p, w_p, ei_n, w_n, npy(xs[0]), npy(xs[1]), params, hop, fill,
                                  directed, npy(xs[2]), npy(xs[3])))
    save(name, edge_index_p=ei_p, edge_weight_p=w_p, edge_index_n=ei_n, edge_weight_n=w_n,
         x_p=npy(xs[0]), x_n=npy(xs[1]), x_pt=npy(xs[2]) if directed else None,
         x_nt=npy(xs[3]) if directed else None, grad_out=npy(go), out=npy(out),
         dx_p=npy(xs[0].grad), dx_n=npy(xs[1].grad),
         dx_pt=npy(xs[2].grad) if directed else None, dx_nt=npy(xs[3].grad) if directed else None,
         hop=np.int64(hop), directed=np.bool_(directed), fill_value=np.float64(fill),
         **{"param" + k: v for k, v in params.items()},
         **{"dparam" + k: npy(p.grad) for k, p in layer.named_parameters()})


def dimpa_case(name, seed, hop, fill=0.5, f=5):
    ei, w = toy_graph(seed)
    n = 40
    g = torch.Generator().manual_seed(seed)
    xs = torch.randn(n, f, generator=g, requires_grad=True)
    xt = torch.randn(n, f, generator=g, requires_grad=True)
    layer = DIMPA(hop, fill)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(torch.rand(p.shape, generator=g) + 0.5)
    out = layer(xs, xt, t(ei), t(w))
    go = torch.randn(out.shape, generator=g)
    (out * go).sum().backward()
    close(name, npy(out), D.dimpa(npy(xs), npy(xt), ei, w, npy(layer._w_s), npy(layer._w_t), hop,
                                  fill))
    save(name, edge_index=ei, edge_weight=w, x_s=npy(xs), x_t=npy(xt), grad_out=npy(go),
         out=npy(out), dx_s=npy(xs.grad), dx_t=npy(xt.grad), w_s=npy(layer._w_s),
         w_t=npy(layer._w_t), dw_s=npy(layer._w_s.grad), dw_t=npy(layer._w_t.grad),
         hop=np.int64(hop), fill_value=np.float64(fill))


# ------------------------------------------------------------------ SGCNConv
def sgcn_case(name, seed, first_aggr, norm_emb, in_dim=6, out_dim=4):
    pos, _ = toy_graph(seed, weighted=False)
    neg, _ = toy_graph(seed + 50, e=100, weighted=False)
    n = 40
    g = torch.Generator().manual_seed(seed)
    fx = in_dim if first_aggr else 2 * in_dim
    x = torch.randn(n, fx, generator=g, requires_grad=True)
    torch.manual_seed(seed)
    layer = SGCNConv(in_dim, out_dim, first_aggr, norm_emb=norm_emb)
    out = layer(x, t(pos), t(neg))
    go = torch.randn(out.shape, generator=g)
    (out * go).sum().backward()
    lb = (npy(layer.lin_b.weight), npy(layer.lin_b.bias))
    lu = (npy(layer.lin_u.weight), npy(layer.lin_u.bias))
    close(name, npy(out), D.sgcn_conv(npy(x), pos, neg, lb, lu, first_aggr, in_dim, norm_emb))
    save(name, pos_edge_index=pos, neg_edge_index=neg, x=npy(x), grad_out=npy(go), out=npy(out),
         dx=npy(x.grad), lin_b_weight=lb[0], lin_b_bias=lb[1], lin_u_weight=lu[0], lin_u_bias=lu[1],
         dlin_b_weight=npy(layer.lin_b.weight.grad), dlin_u_weight=npy(layer.lin_u.weight.grad),
         dlin_b_bias=npy(layer.lin_b.bias.grad), dlin_u_bias=npy(layer.lin_u.bias.grad),
         first_aggr=np.bool_(first_aggr), norm_emb=np.bool_(norm_emb), in_dim=np.int64(in_dim))


def relu_case(name, seed):
    g = torch.Generator().manual_seed(seed)
    re = torch.randn(40, 6, generator=g)
    im = torch.randn(40, 6, generator=g)
    re[3, 2] = 0.0  # boundary: real == 0 passes
    o_r, o_i = complex_relu_layer()(re, im)
    save(name, real=npy(re), imag=npy(im), out_real=npy(o_r), out_imag=npy(o_i))


def kat_case():
    """SURVEY.md Appendix B known-answer vector, re-derived here from the reference and checked
    digit for digit against the hand-checked numbers printed in the survey."""
    ei = np.array([[0, 1, 2, 0, 3], [1, 2, 0, 2, 3]], dtype=np.int64)
    w = np.array([1, 2, 1, 3, 5], dtype=np.float32)
    oei, re, im = get_magnetic_Laplacian(t(ei), t(w), "sym", None, 4, 0.25)
    want_re = [1.128623e-08, 0.7302967, 1.128623e-08, 0.4714045, 0.7302967, 0.4714045, 1, 1, 1, 1]
    want_im = [-0.2581989, 6.384457e-08, 0.2581989, 4.121149e-08, -6.384457e-08, -4.121149e-08,
               0, 0, 0, 0]
    assert npy(oei).tolist() == [[0, 0, 1, 1, 2, 2, 0, 1, 2, 3], [1, 2, 0, 2, 0, 1, 0, 1, 2, 3]]
    assert np.allclose(npy(re), want_re, rtol=2e-6, atol=1e-12)
    assert np.allclose(npy(im), want_im, rtol=2e-6, atol=1e-12)
    layer = MagNetConv(2, 2, 1, 0.25, False)
    with torch.no_grad():
        layer.weight.copy_(torch.tensor([[[1, 0], [0, 1]], [[.5, -1], [2, .25]]]))
        layer.bias.copy_(torch.tensor([.1, -.2]))
    xr = torch.tensor([[1., 2], [3, 4], [5, 6], [7, 8]])
    xi = torch.tensor([[-1, .5], [.25, 2], [1.5, -3], [0, 1]])
    o_r, o_i = layer(xr, xi, t(ei), t(w))
    want_or = [[11.624232, -1.320588], [9.814465, 0.440558], [11.364678, 7.492043], [7.1, 6.8]]
    want_oi = [[11.754375, -0.191489], [10.056267, 3.859610], [14.364678, 1.492042], [7.1, 8.8]]
    assert np.allclose(npy(o_r), want_or, atol=2e-6) and np.allclose(npy(o_i), want_oi, atol=2e-6)
    ws = np.array([1, -2, 1, -3, 5], dtype=np.float32)
    _, sre, sim = get_magnetic_signed_Laplacian(t(ei), t(ws), "sym", None, 4, 0.25)
    save("kat_appendix_b", edge_index=ei, edge_weight=w, lap_index=npy(oei), lap_real=npy(re),
         lap_imag=npy(im), weight=npy(layer.weight), bias=npy(layer.bias), x_real=npy(xr),
         x_imag=npy(xi), out_real=npy(o_r), out_imag=npy(o_i), signed_weight=ws,
         signed_lap_real=npy(sre), signed_lap_imag=npy(sim),
         op_real=npy(layer.cached_result[2]), op_imag=npy(layer.cached_result[3]))


# ------------------------------------------------------------------ attention aggregate (SDGNN / SiGAT)
def gat_cases():
    from torch_geometric.nn import GATConv          # the shim's restatement of PyG GATConv
    from torch_geometric_signed_directed.nn.signed.SDGNN import SDRLayer
    n, f = 40, 6
    ei, _ = toy_graph(71, weighted=False)
    g = torch.Generator().manual_seed(71)
    x = torch.randn(n, f, generator=g, requires_grad=True)
    go = torch.randn(n, 5, generator=g)
    torch.manual_seed(71)
    conv = GATConv(f, 5)
    with torch.no_grad():
        conv.bias.uniform_(-0.5, 0.5)
    out = conv(x, t(ei))
    (out * go).sum().backward()
    close("gat", npy(out), D.gat_conv(npy(x), ei, npy(conv.lin.weight), npy(conv.att_src), npy(conv.att_dst),
                                      npy(conv.bias)))
    save("gat_conv", edge_index=ei, x=npy(x), grad_out=npy(go), out=npy(out), dx=npy(x.grad),
         **{"sd." + k: npy(v) for k, v in conv.state_dict().items()},
         **{"d." + k: npy(p.grad) for k, p in conv.named_parameters()})
    # SDRLayer: the reference's own class over four motif edge lists
    lists = [t(toy_graph(72 + k, e=60 + 20 * k, weighted=False)[0]) for k in range(4)]
    x2 = torch.randn(n, f, generator=g, requires_grad=True)
    torch.manual_seed(72)
    layer = SDRLayer(f, f, edge_lists=lists)
    layer.reset_parameters()
    out2 = layer(x2)
    go2 = torch.randn(out2.shape, generator=g)
    (out2 * go2).sum().backward()
    save("sdr_layer", x=npy(x2), grad_out=npy(go2), out=npy(out2), dx=npy(x2.grad),
         **{f"edges{k}": npy(e) for k, e in enumerate(lists)},
         **{"sd." + k: npy(v) for k, v in layer.state_dict().items()},
         **{"d." + k: npy(p.grad) for k, p in layer.named_parameters()})


# ------------------------------------------------------------------ SSSNET cut objectives (8(f) rank 4)
def loss_case():
    import scipy.sparse as sp
    from torch_geometric_signed_directed.utils.signed import (Prob_Balanced_Normalized_Loss,
                                                              Prob_Balanced_Ratio_Loss, Unhappy_Ratio)
    n, k = 40, 4
    eis, ws = toy_graph(81, signed=True)
    a = sp.coo_matrix((ws, (eis[0], eis[1])), shape=(n, n)).tocsr()
    a_p, a_n = a.maximum(0), (-a).maximum(0)
    a_p.eliminate_zeros()
    a_n.eliminate_zeros()
    g = torch.Generator().manual_seed(81)
    prob = torch.softmax(torch.randn(n, k, generator=g), dim=1).requires_grad_()
    out = {}
    for name, cls in (("normalized", Prob_Balanced_Normalized_Loss), ("ratio", Prob_Balanced_Ratio_Loss),
                      ("unhappy", Unhappy_Ratio)):
        prob.grad = None
        val = cls(a_p, a_n)(prob)
        val.sum().backward()
        out["loss_" + name] = npy(val)
        out["dprob_" + name] = npy(prob.grad)
    # dense float64 check of the normalized loss
    ap, an = a_p.toarray().astype(np.float64), a_n.toarray().astype(np.float64)
    dp, dn = np.diag(ap.sum(1)), np.diag(an.sum(1))
    m, dbar, p = dp - (ap - an), dp + dn, npy(prob).astype(np.float64)
    want = sum(p[:, j] @ m @ p[:, j] / (p[:, j] @ dbar @ p[:, j] + 1e-6) for j in range(k))
    close("loss_normalized", out["loss_normalized"], np.array([want]))
    save("sssnet_losses", edge_index=eis, edge_weight=ws, prob=npy(prob), **out)


def imbalance_loss_case():
    from torch_geometric_signed_directed.utils.directed.prob_imbalance_loss import Prob_Imbalance_Loss
    n, k = 40, 4
    ei, w = toy_graph(91)
    a = torch.sparse_coo_tensor(t(ei), t(w), (n, n)).coalesce()
    g = torch.Generator().manual_seed(91)
    out = {}
    for norm in ("vol_sum", "vol_min", "vol_max", "plain"):
        for thr in ("sort", "std", "naive"):
            prob = torch.softmax(torch.randn(n, k, generator=g), dim=1).requires_grad_()
            val = Prob_Imbalance_Loss(3)(prob, a, k, norm, thr)
            out[f"prob_{norm}_{thr}"] = npy(prob)
            out[f"loss_{norm}_{thr}"] = npy(val)
            if thr == "sort":
                val.sum().backward()
                out[f"dprob_{norm}_{thr}"] = npy(prob.grad)
    save("digrac_imbalance_loss", edge_index=ei, edge_weight=w, **out)


# ------------------------------------------------------------------ model-level callers (eval mode)
def model_case(name, model, args, seed):
    """Reference model in eval mode (dropout off) on fixed inputs: record state_dict + outputs."""
    model.eval()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for prm in model.parameters():               # de-trivialise zero biases / unit weights
            if prm.dim() == 1 and prm.numel() > 1:
                prm.add_(torch.rand(prm.shape, generator=g) - 0.5)
        out = model(*args)
    outs = out if isinstance(out, tuple) else (out,)
    arrs = {"sd." + k: npy(v) for k, v in model.state_dict().items()}
    arrs.update({f"out{i}": npy(o) for i, o in enumerate(outs)})
    return arrs


def models():
    from torch_geometric_signed_directed.nn import DGCN_node_classification
    from torch_geometric_signed_directed.nn import (DIGRAC_node_clustering, DiGCN_Inception_Block_node_classification,
                                                    DiGCN_node_classification, MagNet_link_prediction,
                                                    MagNet_node_classification, MSGNN_link_prediction,
                                                    MSGNN_node_classification, SSSNET_node_clustering)
    n = 40
    ei, w = toy_graph(51)
    eis, ws = toy_graph(52, signed=True)
    g = torch.Generator().manual_seed(50)
    x = torch.randn(n, 6, generator=g)
    xi = torch.randn(n, 6, generator=g)
    query = torch.randint(0, n - 3, (25, 2), generator=g)
    torch.manual_seed(60)
    save("model_magnet_node", edge_index=ei, edge_weight=w, x_real=npy(x), x_imag=npy(xi),
         **model_case("m", MagNet_node_classification(6, hidden=8, q=0.2, K=2, label_dim=4, activation=True,
                                                      layer=2, dropout=0.5), (x, xi, t(ei), t(w)), 61))
    save("model_magnet_link", edge_index=ei, edge_weight=w, x_real=npy(x), x_imag=npy(xi), query=npy(query),
         **model_case("m", MagNet_link_prediction(6, hidden=8, q=0.25, K=1, label_dim=2, layer=2),
                      (x, xi, t(ei), query, t(w)), 62))
    save("model_msgnn_node", edge_index=eis, edge_weight=ws, x_real=npy(x), x_imag=npy(xi),
         **model_case("m", MSGNN_node_classification(6, hidden=8, q=0.1, K=2, label_dim=3, activation=True,
                                                     layer=2, dropout=0.3), (x, xi, t(eis), t(ws)), 63))
    save("model_msgnn_link", edge_index=eis, edge_weight=ws, x_real=npy(x), x_imag=npy(xi), query=npy(query),
         **model_case("m", MSGNN_link_prediction(6, hidden=8, q=0.1, K=2, label_dim=2, layer=2),
                      (x, xi, t(eis), query, t(ws)), 64))
    ei2, w2 = toy_graph(53, e=120)
    save("model_digcn_node", edge_index=ei, edge_weight=w, x=npy(x),
         **model_case("m", DiGCN_node_classification(6, 8, 4, 0.5), (x, t(ei), t(w)), 65))
    save("model_digcn_ib", edge_index=ei, edge_weight=w, edge_index2=ei2, edge_weight2=w2, x=npy(x),
         **model_case("m", DiGCN_Inception_Block_node_classification(6, 8, 4, 0.5),
                      (x, (t(ei), t(ei2)), (t(w), t(w2))), 66))
    save("model_digrac", edge_index=ei, edge_weight=w, x=npy(x),
         **model_case("m", DIGRAC_node_clustering(6, 8, 3, 0.5, 0.5, 2), (t(ei), t(w), x), 67))
    e_in, w_in = toy_graph(55, e=140)
    e_out, w_out = toy_graph(56, e=150)
    for cached in (False, True):
        save("model_dgcn_" + ("cached" if cached else "uncached"), edge_index=ei, edge_in=e_in, w_in=w_in,
             edge_out=e_out, w_out=w_out, x=npy(x), cached=np.bool_(cached),
             **model_case("m", DGCN_node_classification(6, 8, 4, 0.5, improved=True, cached=cached),
                          (x, t(ei), t(e_in), t(e_out), t(w_in), t(w_out)), 69))
    from torch_geometric_signed_directed.nn import (DGCN_link_prediction, DiGCN_Inception_Block_link_prediction,
                                                    DiGCN_link_prediction, SSSNET_link_prediction)
    save("model_digcn_link", edge_index=ei, edge_weight=w, x=npy(x), query=npy(query),
         **model_case("m", DiGCN_link_prediction(6, 8, 2, 0.5), (x, t(ei), query, t(w)), 70))
    save("model_digcn_ib_link", edge_index=ei, edge_weight=w, edge_index2=ei2, edge_weight2=w2, x=npy(x),
         query=npy(query),
         **model_case("m", DiGCN_Inception_Block_link_prediction(6, 8, 3, 0.5),
                      (x, (t(ei), t(ei2)), query, (t(w), t(w2))), 71))
    save("model_dgcn_link", edge_index=ei, edge_in=e_in, w_in=w_in, edge_out=e_out, w_out=w_out, x=npy(x),
         query=npy(query),
         **model_case("m", DGCN_link_prediction(6, 8, 2, 0.5, improved=False, cached=False),
                      (x, t(ei), t(e_in), t(e_out), query, t(w_in), t(w_out)), 72))
    ein, wn = toy_graph(54, e=90)
    for directed in (False, True):
        save("model_sssnet_link_" + ("directed" if directed else "undirected"), edge_index_p=ei, edge_weight_p=w,
             edge_index_n=ein, edge_weight_n=wn, x=npy(x), query=npy(query), directed=np.bool_(directed),
             **model_case("m", SSSNET_link_prediction(6, 8, 3, 0.5, 2, 0.5, directed),
                          (t(ei), t(w), t(ein), t(wn), x, query), 73))
    for directed in (False, True):
        save("model_sssnet_" + ("directed" if directed else "undirected"), edge_index_p=ei, edge_weight_p=w,
             edge_index_n=ein, edge_weight_n=wn, x=npy(x), directed=np.bool_(directed),
             **model_case("m", SSSNET_node_clustering(6, 8, 3, 0.5, 2, 0.5, directed),
                          (t(ei), t(w), t(ein), t(wn), x), 68))


def snea_cases():
    """SNEAConv (first / deep aggregation) with input and parameter gradients, and the SNEA model's z."""
    from torch_geometric_signed_directed.nn.signed.SNEA import SNEA
    from torch_geometric_signed_directed.nn.signed.SNEAConv import SNEAConv
    n = 40
    for name, seed, first in (("snea_first", 91, True), ("snea_deep", 92, False)):
        g = torch.Generator().manual_seed(seed)
        pos = torch.randint(0, n - 3, (2, 110), generator=g)          # nodes 37..39: no positive edge, no loop
        neg = torch.randint(0, n, (2, 70), generator=g)
        pos[:, :4] = pos[0, :4]                                       # listed self loops (dropped, re-added)
        in_dim, out_dim = 5, 4
        x = torch.randn(n, in_dim if first else 2 * in_dim, generator=g, requires_grad=True)
        torch.manual_seed(seed)
        conv = SNEAConv(in_dim, out_dim, first)
        with torch.no_grad():
            for lin in (conv.lin_b, conv.lin_u, conv.alpha_b, conv.alpha_u):
                lin.bias.add_(torch.rand(lin.bias.shape, generator=g) - 0.5)
        out = conv(x, pos, neg)
        pair = lambda l: (l.weight.detach(), l.bias.detach())  # noqa: E731
        want = D.snea_conv(x.detach(), pos, neg, pair(conv.lin_b), pair(conv.lin_u), pair(conv.alpha_b),
                           pair(conv.alpha_u), first, in_dim)
        close(name, out.detach(), want)
        gout = torch.randn(out.shape, generator=g)
        out.backward(gout)
        arrs = {"sd." + k: npy(v) for k, v in conv.state_dict().items()}
        arrs.update({"grad." + k: npy(v.grad) for k, v in conv.named_parameters()})
        save(name, pos=npy(pos), neg=npy(neg), x=npy(x), out=npy(out), gout=npy(gout), dx=npy(x.grad),
             dense_f64=want, first_aggr=np.bool_(first), **arrs)
    g = torch.Generator().manual_seed(93)
    pairs = torch.randint(0, n, (2, 150), generator=g)
    sign = torch.where(torch.rand(150, generator=g) < 0.6, 1, -1)
    edge_index_s = torch.cat([pairs.t(), sign[:, None]], dim=1)
    init = torch.randn(n, 6, generator=g)
    torch.manual_seed(94)
    model = SNEA(n, edge_index_s, in_dim=6, out_dim=8, layer_num=3, init_emb=init)
    save("model_snea", edge_index_s=npy(edge_index_s), init_emb=npy(init), z=npy(model()),
         **{"sd." + k: npy(v) for k, v in model.state_dict().items()})


def sdgnn_case():
    """SDGNN: embeddings through two SDRLayers (4 GATConv aggregators each) and its three objectives (all
    deterministic: no negative sampling), plus the motif-count matrix the triangle loss weights edges with."""
    from torch_geometric_signed_directed.nn.signed.SDGNN import SDGNN
    n = 40
    g = torch.Generator().manual_seed(95)
    pairs = torch.randint(0, n, (170, 2), generator=g)
    pairs = pairs[pairs[:, 0] != pairs[:, 1]]
    sign = torch.where(torch.rand(pairs.size(0), generator=g) < 0.6, 1, -1)
    edge_index_s = torch.cat([pairs, sign[:, None]], dim=1)
    edge_index_s = torch.cat([edge_index_s, edge_index_s[:6] * torch.tensor([1, 1, -1]), edge_index_s[6:12]])
    init = torch.randn(n, 8, generator=g)
    torch.manual_seed(96)
    model = SDGNN(n, edge_index_s, in_dim=8, out_dim=8, layer_num=2, init_emb=init)
    with torch.no_grad():
        for prm in model.parameters():
            if prm.dim() == 1 and prm.numel() > 1:
                prm.add_(torch.rand(prm.shape, generator=g) - 0.5)
    z = model()
    pos, neg = model.pos_edge_index, model.neg_edge_index
    tri = model.tri_weight.tocoo()
    save("model_sdgnn", edge_index_s=npy(edge_index_s), init_emb=npy(init), z=npy(z),
         loss_sign=npy(model.loss_sign(z, pos, neg)), loss_direction=npy(model.loss_direction(z, pos, neg)),
         loss_tri=npy(model.loss_tri(z, pos, neg)), loss_total=npy(model.loss()),
         tri_row=tri.row.astype(np.int64), tri_col=tri.col.astype(np.int64), tri_val=tri.data.astype(np.int64),
         **{"sd." + k: npy(v) for k, v in model.state_dict().items()})


def sigat_case():
    """SiGAT: 38 GATConv aggregators over the motif neighbourhoods + MLP, link-sign product loss."""
    from torch_geometric_signed_directed.nn.signed.SiGAT import SiGAT
    n = 40
    g = torch.Generator().manual_seed(97)
    pairs = torch.randint(0, n, (420, 2), generator=g)           # dense enough that every motif list is non-empty
    pairs = pairs[pairs[:, 0] != pairs[:, 1]]
    sign = torch.where(torch.rand(pairs.size(0), generator=g) < 0.55, 1, -1)
    edge_index_s = torch.cat([pairs, sign[:, None]], dim=1)
    init = torch.randn(n, 8, generator=g)
    torch.manual_seed(98)
    model = SiGAT(n, edge_index_s, in_dim=8, out_dim=8, init_emb=init)
    assert all(e.dim() == 2 and e.size(1) > 0 for e in model.edge_lists)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if prm.dim() == 1 and prm.numel() > 1:
                prm.add_(torch.rand(prm.shape, generator=g) - 0.5)
    z = model()
    save("model_sigat", edge_index_s=npy(edge_index_s), init_emb=npy(init), z=npy(z), loss=npy(model.loss()),
         list_sizes=np.array([e.size(1) for e in model.edge_lists]),
         **{"sd." + k: npy(v) for k, v in model.state_dict().items()})


def digcl_case():
    """DiGCL: 3-layer GCNConv encoder (prelu) on two weighted views, projection head, both contrastive losses."""
    from torch_geometric_signed_directed.nn.directed.DiGCL import DiGCL
    n = 40
    ei, w = toy_graph(101)
    ei2, w2 = toy_graph(102, e=120)
    g = torch.Generator().manual_seed(103)
    x1, x2 = torch.randn(n, 6, generator=g), torch.randn(n, 6, generator=g)
    torch.manual_seed(104)
    model = DiGCL(6, 'prelu', 8, 5, 0.4, 3)
    with torch.no_grad():
        for prm in model.parameters():
            if prm.dim() == 1 and prm.numel() > 1:
                prm.add_(torch.rand(prm.shape, generator=g) - 0.5)
    model.eval()
    z1 = model(x1, t(ei), t(w))
    z2 = model(x2, t(ei2), t(w2))
    z3 = model(x1, t(ei))                                   # unweighted view
    # independent dense check of one GCNConv layer
    conv = model.encoder.conv[0]
    a = D._with_remaining_loops(ei, w, n, 1.0)
    dis = D._inv_pow(a.sum(0), -0.5)
    want = (dis[:, None] * a * dis[None, :]).T @ (x1.double().numpy() @ conv.lin.weight.detach().double().numpy().T) \
        + conv.bias.detach().double().numpy()
    close("gcnconv", conv(x1, t(ei), t(w)).detach(), want)
    save("model_digcl", edge_index=ei, edge_weight=w, edge_index2=ei2, edge_weight2=w2, x1=npy(x1), x2=npy(x2),
         z1=npy(z1), z2=npy(z2), z3=npy(z3), loss=npy(model.loss(z1, z2)), loss_sum=npy(model.loss(z1, z2, mean=False)),
         loss_batched=npy(model.loss(z1, z2, batch_size=16)),
         **{"sd." + k: npy(v) for k, v in model.state_dict().items()})


def digcn_adjs_case():
    """DiGCN / DiGCL operator pre-processing (utils/directed/get_adjs_DiGCN.py) on a toy graph."""
    import torch_geometric_signed_directed.utils.directed.get_adjs_DiGCN as A
    n = 40
    ei, w = toy_graph(111)
    out = {}
    for name, res in (("second", A.get_second_directed_adj(t(ei), n, torch.float32, t(w))),
                      ("second_unw", A.get_second_directed_adj(t(ei), n, torch.float32, None)),
                      ("appr", A.get_appr_directed_adj(0.1, t(ei), n, torch.float32, t(w))),
                      ("appr_unw", A.get_appr_directed_adj(0.2, t(ei), n, torch.float32, None)),
                      ("fast", A.cal_fast_appr(0.1, t(ei), n, torch.float32, t(w)))):
        out[name + "_index"], out[name + "_value"] = npy(res[0]), npy(res[1])
    save("adjs_digcn", edge_index=ei, edge_weight=w, **out)


def sgcn_model_and_sign_losses():
    """SGCN.forward (z) with given initial embeddings, and the signed objectives with the random negative
    draws of PyG replaced by fixed index sets (patched into the reference module), so the arithmetic is pinned."""
    import torch_geometric_signed_directed.utils.signed.link_sign_loss as L
    from torch_geometric_signed_directed.nn.signed.SGCN import SGCN
    from torch_geometric_signed_directed.utils.signed import create_spectral_features
    n = 40
    g = torch.Generator().manual_seed(80)
    pairs = torch.randint(0, n, (2, 150), generator=g)
    pairs = pairs[:, pairs[0] != pairs[1]]
    sign = torch.where(torch.rand(pairs.size(1), generator=g) < 0.6, 1, -1)
    edge_index_s = torch.cat([pairs.t(), sign[:, None]], dim=1)
    init = torch.randn(n, 6, generator=g)
    torch.manual_seed(81)
    model = SGCN(n, edge_index_s, in_dim=6, out_dim=8, layer_num=3, init_emb=init, norm_emb=True)
    with torch.no_grad():
        for prm in model.parameters():
            if prm.dim() == 1 and prm.numel() > 1:
                prm.add_(torch.rand(prm.shape, generator=g) - 0.5)
    z = model()
    pos, neg = model.pos_edge_index, model.neg_edge_index
    none_ei = torch.randint(0, n, (2, 70), generator=g)
    k_pos = torch.randint(0, n, (pos.size(1),), generator=g)
    k_neg = torch.randint(0, n, (neg.size(1),), generator=g)
    L.negative_sampling = lambda ei, num: none_ei
    ks = {pos.size(1): k_pos, neg.size(1): k_neg}
    assert pos.size(1) != neg.size(1)
    L.structured_negative_sampling = lambda ei, num: (ei[0], ei[1], ks[ei.size(1)])
    entropy = model.lsp_loss(z, pos, neg)
    structure = model.structure_loss(z, pos, neg)
    total = model.loss()
    close("sgcn.loss", total.detach(), (entropy + model.lamb * structure).detach())
    torch.manual_seed(82)
    direction = L.Sign_Direction_Loss(8)
    arrs = {"sd." + k_: npy(v) for k_, v in model.state_dict().items()}
    arrs.update({"dir." + k_: npy(v) for k_, v in direction.state_dict().items()})
    feats = create_spectral_features(pos, neg, n, 5)
    save("model_sgcn", edge_index_s=npy(edge_index_s), init_emb=npy(init), z=npy(z), none_edge_index=npy(none_ei),
         k_pos=npy(k_pos), k_neg=npy(k_neg), loss_entropy=npy(entropy), loss_structure=npy(structure),
         loss_total=npy(total), loss_product=npy(L.Link_Sign_Product_Loss()(z, pos, neg)),
         loss_product_entropy=npy(L.Sign_Product_Entropy_Loss()(z, pos, neg)),
         loss_direction=npy(direction(z, pos, neg)), spectral=npy(feats), **arrs)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    print("MagNetConv / MSConv")
    magnet_case("magnet_k1_sym_w", 1, 1, "sym", True)
    magnet_case("magnet_k1_sym_unw", 2, 1, "sym", False)
    magnet_case("magnet_k2_sym_w", 3, 2, "sym", True, q=0.1)
    magnet_case("magnet_k3_sym_w_nobias", 4, 3, "sym", True, bias=False)
    magnet_case("magnet_k2_none_w", 5, 2, None, True, q=0.2)
    magnet_case("msconv_k1_sym_abs", 6, 1, "sym", True, signed=True)
    magnet_case("msconv_k2_sym_noabs", 7, 2, "sym", True, signed=True, absolute_degree=False)
    magnet_case("msconv_k2_none_abs", 8, 2, None, True, signed=True, q=0.15)
    magnet_case("magnet_k1_sym_wide", 9, 1, "sym", True, fin=16, fout=4)
    magnet_case("magnet_k3_sym_wide", 10, 3, "sym", True, fin=16, fout=4, q=0.2)
    magnet_case("msconv_k2_sym_wide", 19, 2, "sym", True, signed=True, fin=12, fout=5)
    print("DiGCNConv / DGCNConv / Conv_Base")
    digcn_case("digcn_bias", 11)
    digcn_case("digcn_nobias", 12, bias=False)
    dgcn_case("dgcn_w_improved", 13, True, True, True)
    dgcn_case("dgcn_unw", 14, False, False, True)
    dgcn_case("dgcn_w_noloops", 15, True, False, False)
    conv_base_case("conv_base_w_fill05", 16, True, 0.5)
    conv_base_case("conv_base_unw_fill0", 17, False, 0.0)
    print("SIMPA / DIMPA")
    simpa_case("simpa_undirected_hop2", 21, 2, False)
    simpa_case("simpa_undirected_hop3", 22, 3, False)
    simpa_case("simpa_directed_hop2", 23, 2, True)
    dimpa_case("dimpa_hop2", 24, 2)
    print("SGCNConv")
    sgcn_case("sgcn_first", 31, True, False)
    sgcn_case("sgcn_deep", 32, False, False)
    sgcn_case("sgcn_first_normemb", 33, True, True)
    relu_case("complex_relu", 41)
    kat_case()
    print("attention aggregate")
    gat_cases()
    print("SSSNET cut objectives / DIGRAC imbalance")
    loss_case()
    imbalance_loss_case()
    print("model-level callers")
    models()
    sgcn_model_and_sign_losses()
    snea_cases()
    sdgnn_case()
    sigat_case()
    digcl_case()
    digcn_adjs_case()


if __name__ == "__main__":
    main()
