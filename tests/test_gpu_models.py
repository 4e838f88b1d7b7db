"""GPU: the thin model callers (nn/models.py) loaded with the REFERENCE's state_dict and run in eval
mode must reproduce the reference's recorded outputs (tests/golden/model_*.npz) within 1e-5."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from tolerance import close

pytestmark = pytest.mark.gpu
D = "cuda:0"


def load(model, g):
    sd = {k[3:]: g.t(k) for k in g if k.startswith("sd.")}
    model.load_state_dict(sd, strict=True)      # identical key set => reference checkpoints load
    return model.to(D).eval()


def outs(g):
    return [g[k] for k in sorted(k for k in g if k.startswith("out") and k[3:].isdigit())]


def check(result, g):
    result = result if isinstance(result, tuple) else (result,)
    want = outs(g)
    assert len(result) == len(want)
    for r, w in zip(result, want):
        if w.dtype.kind in "iu":
            assert np.array_equal(r.cpu().numpy(), w)      # argmax cluster predictions: exact
        else:
            close(r, w)


def test_magnet_models():
    from pytorch_geometric_signed_directed_amd.nn import MagNet_link_prediction, MagNet_node_classification
    g = load_golden("model_magnet_node")
    m = load(MagNet_node_classification(6, hidden=8, q=0.2, K=2, label_dim=4, activation=True, layer=2, dropout=0.5), g)
    with torch.no_grad():
        check(m(g.t("x_real", D), g.t("x_imag", D), g.t("edge_index", D), g.t("edge_weight", D)), g)
    g = load_golden("model_magnet_link")
    m = load(MagNet_link_prediction(6, hidden=8, q=0.25, K=1, label_dim=2, layer=2), g)
    with torch.no_grad():
        check(m(g.t("x_real", D), g.t("x_imag", D), g.t("edge_index", D), g.t("query", D), g.t("edge_weight", D)), g)


def test_msgnn_models():
    from pytorch_geometric_signed_directed_amd.nn import MSGNN_link_prediction, MSGNN_node_classification
    g = load_golden("model_msgnn_node")
    m = load(MSGNN_node_classification(6, hidden=8, q=0.1, K=2, label_dim=3, activation=True, layer=2, dropout=0.3), g)
    with torch.no_grad():
        check(m(g.t("x_real", D), g.t("x_imag", D), g.t("edge_index", D), g.t("edge_weight", D)), g)
    g = load_golden("model_msgnn_link")
    m = load(MSGNN_link_prediction(6, hidden=8, q=0.1, K=2, label_dim=2, layer=2), g)
    with torch.no_grad():
        check(m(g.t("x_real", D), g.t("x_imag", D), g.t("edge_index", D), g.t("query", D), g.t("edge_weight", D)), g)


def test_digcn_models():
    from pytorch_geometric_signed_directed_amd.nn import (DiGCN_Inception_Block_node_classification,
                                                          DiGCN_node_classification)
    g = load_golden("model_digcn_node")
    m = load(DiGCN_node_classification(6, 8, 4, 0.5), g)
    with torch.no_grad():
        check(m(g.t("x", D), g.t("edge_index", D), g.t("edge_weight", D)), g)
    g = load_golden("model_digcn_ib")
    m = load(DiGCN_Inception_Block_node_classification(6, 8, 4, 0.5), g)
    with torch.no_grad():
        check(m(g.t("x", D), (g.t("edge_index", D), g.t("edge_index2", D)),
                (g.t("edge_weight", D), g.t("edge_weight2", D))), g)


def test_clustering_models():
    from pytorch_geometric_signed_directed_amd.nn import DIGRAC_node_clustering, SSSNET_node_clustering
    g = load_golden("model_digrac")
    m = load(DIGRAC_node_clustering(6, 8, 3, 0.5, 0.5, 2), g)
    with torch.no_grad():
        check(m(g.t("edge_index", D), g.t("edge_weight", D), g.t("x", D)), g)
    for name in ("model_sssnet_undirected", "model_sssnet_directed"):
        g = load_golden(name)
        m = load(SSSNET_node_clustering(6, 8, 3, 0.5, 2, 0.5, bool(g["directed"])), g)
        with torch.no_grad():
            check(m(g.t("edge_index_p", D), g.t("edge_weight_p", D), g.t("edge_index_n", D),
                    g.t("edge_weight_n", D), g.t("x", D)), g)


def test_model_trains_one_step():
    """End-to-end: loss.backward() + optimiser step through the fused layer path (h=16)."""
    from pytorch_geometric_signed_directed_amd.nn import MagNet_node_classification
    g = load_golden("model_magnet_node")
    torch.manual_seed(0)
    m = MagNet_node_classification(6, hidden=16, K=1, label_dim=4, activation=True, layer=2, dropout=0.0, cached=True).to(D)
    opt = torch.optim.Adam(m.parameters(), lr=0.01)
    y = torch.randint(0, 4, (40,), device=D)
    args = (g.t("x_real", D), g.t("x_imag", D), g.t("edge_index", D), g.t("edge_weight", D))
    losses = []
    for _ in range(5):
        opt.zero_grad()
        loss = torch.nn.functional.nll_loss(m(*args), y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0]


def test_train_step_captures_into_a_hipgraph():
    """The C-ABI kernels are plain stream launches, so a whole cached-operator train step can be captured
    by torch.cuda.graphs (hipStreamBeginCapture) and replayed: replay must reproduce the eager result."""
    from pytorch_geometric_signed_directed_amd.nn import MagNetConv
    g = load_golden("model_magnet_node")
    torch.manual_seed(0)
    layer = MagNetConv(6, 16, 2, 0.25, False, cached=True).to(D)
    xr, xi = g.t("x_real", D), g.t("x_imag", D)
    ei, w = g.t("edge_index", D), g.t("edge_weight", D)
    a, b = xr.clone().requires_grad_(), xi.clone().requires_grad_()

    def step():
        layer.zero_grad(set_to_none=False)
        if a.grad is not None:
            a.grad.zero_()
        o = layer(a, b, ei, w)
        (o[0].square().sum() + o[1].sum()).backward()
        return o

    from pytorch_geometric_signed_directed_amd.hipgraph import capture_step
    for _ in range(2):
        step()
    want_o = [t.detach().clone() for t in step()]
    want_g = layer.weight.grad.detach().clone()
    replay = capture_step(step, warmup=3)
    layer.weight.grad.zero_()
    static_o = replay()
    torch.cuda.synchronize()
    assert torch.equal(static_o[0], want_o[0]) and torch.equal(static_o[1], want_o[1])
    assert torch.equal(layer.weight.grad, want_g)


def test_training_trajectory_matches_the_oracle_model():
    """Five Adam steps of MagNet_node_classification (2 layers, h=16 -> the fused MFMA path, K=2,
    complex ReLU) on the GPU against the same model assembled from the ORACLE's reference op sequence on
    the CPU, same initial weights: the loss curves must track each other (fp32 drift only)."""
    from oracle import ref_layers as R
    from pytorch_geometric_signed_directed_amd.nn import MagNet_node_classification
    g = torch.Generator().manual_seed(21)
    n, e, f, h, c = 3000, 40000, 32, 16, 5
    ei = torch.randint(0, n, (2, e), generator=g)
    w = torch.rand(e, generator=g) + 0.5
    x = torch.randn(n, f, generator=g)
    y = torch.randint(0, c, (n,), generator=g)
    torch.manual_seed(21)
    model = MagNet_node_classification(f, hidden=h, K=2, label_dim=c, activation=True, layer=2, dropout=0.0, cached=True)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    # oracle model: explicit parameter tensors + reference op sequence
    prm = {k: v.clone().requires_grad_() for k, v in sd.items()}
    op = R.magnet_operator(ei, w, n, 0.25, "sym", 2.0)

    def oracle_forward():
        re, im = x, x
        for layer in range(2):
            re, im = R.magnet_conv(re, im, op, prm[f"Chebs.{layer}.weight"], prm[f"Chebs.{layer}.bias"], duplicate=False)
            re, im = R.complex_relu(re, im)
        z = torch.cat([re, im], dim=-1)
        logits = torch.nn.functional.conv1d(z.t().unsqueeze(0), prm["Conv.weight"], prm["Conv.bias"])
        return torch.log_softmax(logits, dim=1)[0].t()

    opt_o = torch.optim.Adam(list(prm.values()), lr=0.01)
    model.to(D)
    opt_g = torch.optim.Adam(model.parameters(), lr=0.01)
    xd, eid, wd, yd = x.to(D), ei.to(D), w.to(D), y.to(D)
    for step in range(5):
        opt_o.zero_grad()
        lo = torch.nn.functional.nll_loss(oracle_forward(), y)
        lo.backward()
        opt_o.step()
        opt_g.zero_grad()
        lg = torch.nn.functional.nll_loss(model(xd, xd, eid, wd), yd)
        lg.backward()
        opt_g.step()
        assert abs(float(lg) - float(lo)) <= 2e-4 * max(1.0, abs(float(lo))), (step, float(lg), float(lo))
    # after training, individual logits may differ where a complex-ReLU mask sits on its discontinuity
    # (real ~ 0) or Adam normalised a near-zero gradient; the bulk must still agree
    with torch.no_grad():
        diff = (model(xd, xd, eid, wd).cpu() - oracle_forward().detach()).abs()
    assert float(diff.mean()) <= 5e-3 and float(diff.median()) <= 5e-3


def test_dgcn_model_incl_shared_cached_conv_quirk():
    from pytorch_geometric_signed_directed_amd.nn import DGCN_node_classification
    for name in ("model_dgcn_uncached", "model_dgcn_cached"):
        g = load_golden(name)
        m = load(DGCN_node_classification(6, 8, 4, 0.5, improved=True, cached=bool(g["cached"])), g)
        with torch.no_grad():
            check(m(g.t("x", D), g.t("edge_index", D), g.t("edge_in", D), g.t("edge_out", D), g.t("w_in", D),
                    g.t("w_out", D)), g)


def test_link_prediction_models():
    from pytorch_geometric_signed_directed_amd.nn import (DGCN_link_prediction, DiGCN_Inception_Block_link_prediction,
                                                          DiGCN_link_prediction, SSSNET_link_prediction)
    g = load_golden("model_digcn_link")
    m = load(DiGCN_link_prediction(6, 8, 2, 0.5), g)
    with torch.no_grad():
        check(m(g.t("x", D), g.t("edge_index", D), g.t("query", D), g.t("edge_weight", D)), g)
    g = load_golden("model_digcn_ib_link")
    m = load(DiGCN_Inception_Block_link_prediction(6, 8, 3, 0.5), g)
    with torch.no_grad():
        check(m(g.t("x", D), (g.t("edge_index", D), g.t("edge_index2", D)), g.t("query", D),
                (g.t("edge_weight", D), g.t("edge_weight2", D))), g)
    g = load_golden("model_dgcn_link")
    m = load(DGCN_link_prediction(6, 8, 2, 0.5, improved=False, cached=False), g)
    with torch.no_grad():
        check(m(g.t("x", D), g.t("edge_index", D), g.t("edge_in", D), g.t("edge_out", D), g.t("query", D),
                g.t("w_in", D), g.t("w_out", D)), g)
    for name in ("model_sssnet_link_undirected", "model_sssnet_link_directed"):
        g = load_golden(name)
        m = load(SSSNET_link_prediction(6, 8, 3, 0.5, 2, 0.5, bool(g["directed"])), g)
        with torch.no_grad():
            check(m(g.t("edge_index_p", D), g.t("edge_weight_p", D), g.t("edge_index_n", D),
                    g.t("edge_weight_n", D), g.t("x", D), g.t("query", D)), g)


def test_sgcn_model_and_signed_objectives():
    """SGCN.forward() (3 SGCNConv layers, tanh, normalised embeddings) against the reference's z, and its
    objectives with the reference's random negative draws replaced by the recorded index sets."""
    from pytorch_geometric_signed_directed_amd.nn import SGCN
    from pytorch_geometric_signed_directed_amd.utils.signed import (Link_Sign_Product_Loss, Sign_Direction_Loss,
                                                                     Sign_Product_Entropy_Loss)
    g = load_golden("model_sgcn")
    m = SGCN(40, g.t("edge_index_s"), in_dim=6, out_dim=8, layer_num=3, init_emb=g.t("init_emb"), norm_emb=True)
    m = load(m, g)
    assert m.pos_edge_index.is_cuda and m.x.is_cuda
    z = m()
    close(z, g["z"])
    pos, neg = m.pos_edge_index, m.neg_edge_index
    close(m.lsp_loss(z, pos, neg, g.t("none_edge_index", D)), g["loss_entropy"])
    st = m.structure_loss
    close(st.pos_embedding_loss(z, pos, g.t("k_pos", D)) + st.neg_embedding_loss(z, neg, g.t("k_neg", D)),
          g["loss_structure"])
    close(Link_Sign_Product_Loss()(z, pos, neg), g["loss_product"])
    close(Sign_Product_Entropy_Loss()(z, pos, neg), g["loss_product_entropy"])
    d = Sign_Direction_Loss(8)
    d.load_state_dict({k[4:]: g.t(k) for k in g if k.startswith("dir.")})
    close(d.to(D)(z, pos, neg), g["loss_direction"])
    loss = m.loss()                                   # sampled negatives: finite, differentiable
    loss.backward()
    assert torch.isfinite(loss) and m.conv1.lin_b.weight.grad.abs().sum() > 0


def test_snea_model():
    from pytorch_geometric_signed_directed_amd.nn import SNEA
    g = load_golden("model_snea")
    m = load(SNEA(40, g.t("edge_index_s"), in_dim=6, out_dim=8, layer_num=3, init_emb=g.t("init_emb")), g)
    z = m()
    close(z, g["z"])
    m.loss().backward()
    assert m.x.grad is not None and torch.isfinite(m.x.grad).all()        # init_emb_grad defaults to True


def test_sdgnn_model():
    """SDGNN: z through two SDRLayers and the three deterministic objectives against the reference; the motif
    weights (sparse-product restatement of the reference's set loops) must equal its matrix entry for entry."""
    import scipy.sparse as sp
    from pytorch_geometric_signed_directed_amd.nn import SDGNN
    g = load_golden("model_sdgnn")
    m = SDGNN(40, g.t("edge_index_s"), in_dim=8, out_dim=8, layer_num=2, init_emb=g.t("init_emb"))
    want = sp.coo_matrix((g["tri_val"], (g["tri_row"], g["tri_col"])), shape=(40, 40)).tocsr()
    assert abs(m.tri_weight.tocsr() - want).sum() == 0
    m = load(m, g)
    assert all(e.is_cuda for e in m.edge_lists) and m.layers[0].edge_lists is m.edge_lists
    z = m()
    close(z, g["z"])
    pos, neg = m.pos_edge_index, m.neg_edge_index
    close(m.loss_sign(z, pos, neg), g["loss_sign"])
    close(m.loss_direction(z, pos, neg), g["loss_direction"])
    close(m.loss_tri(z, pos, neg), g["loss_tri"])
    loss = m.loss()
    close(loss, g["loss_total"])
    loss.backward()
    assert torch.isfinite(m.x.grad).all() and m.x.grad.abs().sum() > 0


def test_sigat_model():
    """SiGAT: 38 motif neighbourhoods (sizes must equal the reference's), 38 GATConv aggregates + MLP, product loss."""
    from pytorch_geometric_signed_directed_amd.nn import SiGAT
    g = load_golden("model_sigat")
    m = SiGAT(40, g.t("edge_index_s"), in_dim=8, out_dim=8, init_emb=g.t("init_emb"))
    assert [e.size(1) for e in m.edge_lists] == g["list_sizes"].tolist()
    m = load(m, g)
    z = m()
    close(z, g["z"])
    loss = m.loss()
    close(loss, g["loss"])
    loss.backward()
    assert torch.isfinite(m.x.grad).all() and m.agg_37.lin.weight.grad.abs().sum() > 0


def test_digcl_model():
    """DiGCL: GCNConv encoder (weighted and unweighted views), projection head and both contrastive losses."""
    from pytorch_geometric_signed_directed_amd.nn.directed import DiGCL
    g = load_golden("model_digcl")
    m = load(DiGCL(6, 'prelu', 8, 5, 0.4, 3), g)
    z1 = m(g.t("x1", D), g.t("edge_index", D), g.t("edge_weight", D))
    z2 = m(g.t("x2", D), g.t("edge_index2", D), g.t("edge_weight2", D))
    close(z1, g["z1"]); close(z2, g["z2"])
    close(m(g.t("x1", D), g.t("edge_index", D)), g["z3"])
    close(m.loss(z1, z2), g["loss"])
    close(m.loss(z1, z2, mean=False), g["loss_sum"])
    close(m.loss(z1, z2, batch_size=16), g["loss_batched"])
    m.loss(z1, z2).backward()
    assert m.encoder.conv[0].lin.weight.grad.abs().sum() > 0


def test_c1_magnet_node_classification_in_its_stated_shape():
    """BASELINE configs[0] as stated (examples/magnet_node.py:22-29,56-62; MagNet_node_classification.py:66-92):
    MagNet_node_classification(hidden=16, K=1, layer=2, activation off, dropout off) with `cached=False` (the operator
    is rebuilt by BOTH layers of every forward -- the fused build), the raw 2879-wide first layer (Clenshaw route) and
    X_img = X_real (one leaf feeding both inputs), on a DSBM of Cora-ML size (cora_ml.npz itself is not shipped:
    N = 2995, E = 8416, 7 classes, sparse row-normalised bag-of-words features).  Train-mode forward, NLL on a train
    mask, backward: log-probabilities and the gradient of every parameter and of the input against the oracle's
    reference op sequence."""
    import torch.nn.functional as F
    from oracle import ref_layers as R
    from pytorch_geometric_signed_directed_amd import graphs
    from pytorch_geometric_signed_directed_amd.nn import MagNet_node_classification
    n, e, f_in, classes, hidden = 2995, 8416, 2879, 7, 16
    ei_np, labels, _ = graphs.dsbm_for_edges(n, e, k=classes, seed=21)
    ei = torch.from_numpy(ei_np)
    g = torch.Generator().manual_seed(21)
    x = (torch.rand(n, f_in, generator=g) < 0.0175).float()
    x = x / x.sum(1, keepdim=True).clamp(min=1.0)
    y = torch.from_numpy(labels)
    mask = torch.rand(n, generator=g) < 0.2
    torch.manual_seed(21)
    model = MagNet_node_classification(q=0.25, K=1, num_features=f_in, hidden=hidden, label_dim=classes)
    assert not model.Chebs[0].cached and not model.activation and not model.dropout
    with torch.no_grad():
        for cheb in model.Chebs:
            cheb.bias.uniform_(-0.2, 0.2)
    # oracle (CPU, reference op sequence)
    prm = {k: v.detach().clone().requires_grad_() for k, v in model.named_parameters()}
    xo = x.clone().requires_grad_()
    op = R.magnet_operator(ei, None, n, 0.25, "sym", 2.0)
    r1, i1 = R.magnet_conv(xo, xo, op, prm["Chebs.0.weight"], prm["Chebs.0.bias"], duplicate=False)
    r2, i2 = R.magnet_conv(r1, i1, op, prm["Chebs.1.weight"], prm["Chebs.1.bias"], duplicate=False)
    logits = torch.cat((r2, i2), dim=-1) @ prm["Conv.weight"][:, :, 0].t() + prm["Conv.bias"]
    want = F.log_softmax(logits, dim=1)
    F.nll_loss(want[mask], y[mask]).backward()
    # the layers on the GPU
    model.to(D).train()
    xd = x.to(D).requires_grad_()
    eid = ei.to(D)
    got = model(xd, xd, edge_index=eid, edge_weight=None)
    F.nll_loss(got[mask.to(D)], y.to(D)[mask.to(D)]).backward()
    close(got, want.detach(), what="C1 log-probabilities")
    close(xd.grad, xo.grad, what="C1 d loss / d X (real and imaginary inputs share the leaf)")
    for name, p in model.named_parameters():
        close(p.grad, prm[name].grad, norm=True, what=f"C1 d {name}")
    # cached=False: both layers built their operator during the forward, each through the fused build
    assert all(cheb._operator is not None and cheb._operator._off_index is None for cheb in model.Chebs)
