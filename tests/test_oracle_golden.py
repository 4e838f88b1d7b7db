"""CPU: the oracle (oracle/ref_layers.py, the restatement) against the golden vectors recorded
from the reference's own Python (oracle/gen_golden.py) and against the hand-checked KAT of
SURVEY.md Appendix B.  Tolerance: 1e-6 * scale (fp32; same ATen op sequence => usually exact)."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from oracle import ref_layers as R

TOL = 1e-6


def assert_close(got, want, tol=TOL):
    got = got.detach().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape
    scale = max(1.0, float(np.abs(want).max()))
    assert float(np.abs(got.astype(np.float64) - want).max()) <= tol * scale


def _magnet(g):
    n = g["x_real"].shape[0]
    norm = None if str(g["normalization"]) == "none" else "sym"
    lam = float(g["lambda_max"]) if "lambda_max" in g else 2.0
    return R.magnet_operator(g.t("edge_index"), g.t("edge_weight"), n, float(g["q"]), norm, lam,
                             bool(g["signed"]), bool(g["absolute_degree"]))


@pytest.mark.parametrize("name", golden_names("magnet_") + golden_names("msconv_"))
def test_magnet_msconv(name):
    g = load_golden(name)
    op = _magnet(g)
    assert op[0].tolist() == g["op_index_real"].tolist()
    assert op[1].tolist() == g["op_index_imag"].tolist()
    assert_close(op[2], g["op_real"])
    assert_close(op[3], g["op_imag"])
    xr, xi = g.t("x_real").requires_grad_(), g.t("x_imag").requires_grad_()
    w = g.t("weight").requires_grad_()
    b = g.t("bias")
    if b is not None:
        b.requires_grad_()
    o_r, o_i = R.magnet_conv(xr, xi, op, w, b)
    assert_close(o_r, g["out_real"])
    assert_close(o_i, g["out_imag"])
    ((o_r * g.t("grad_real")).sum() + (o_i * g.t("grad_imag")).sum()).backward()
    assert_close(xr.grad, g["dx_real"])
    assert_close(xi.grad, g["dx_imag"])
    assert_close(w.grad, g["dweight"], 2e-6)
    if b is not None:
        assert_close(b.grad, g["dbias"], 2e-6)
    # the de-duplicated evaluation is the same function
    o_r2, o_i2 = R.magnet_conv(xr, xi, op, w, b, duplicate=False)
    assert torch.equal(o_r2, o_r) and torch.equal(o_i2, o_i)


def test_lambda_max_matches_reference():
    g = load_golden("magnet_k2_none_w")
    lam = R.laplacian_lambda_max(g.t("edge_index"), g.t("edge_weight"), 40, float(g["q"]))
    assert abs(lam - float(g["lambda_max"])) <= 1e-5 * float(g["lambda_max"])
    g = load_golden("msconv_k2_none_abs")
    lam = R.laplacian_lambda_max(g.t("edge_index"), g.t("edge_weight"), 40, float(g["q"]), True, True)
    assert abs(lam - float(g["lambda_max"])) <= 1e-5 * float(g["lambda_max"])


def test_kat_appendix_b():
    g = load_golden("kat_appendix_b")
    ei, re, im = R.magnetic_laplacian(g.t("edge_index"), g.t("edge_weight"), 4, 0.25)
    assert ei.tolist() == [[0, 0, 1, 1, 2, 2, 0, 1, 2, 3], [1, 2, 0, 2, 0, 1, 0, 1, 2, 3]]
    np.testing.assert_allclose(re.numpy(), [1.128623e-08, 0.7302967, 1.128623e-08, 0.4714045,
                                            0.7302967, 0.4714045, 1, 1, 1, 1], rtol=2e-6, atol=1e-12)
    np.testing.assert_allclose(im.numpy(), [-0.2581989, 6.384457e-08, 0.2581989, 4.121149e-08,
                                            -6.384457e-08, -4.121149e-08, 0, 0, 0, 0],
                               rtol=2e-6, atol=1e-12)
    op = R.magnet_operator(g.t("edge_index"), g.t("edge_weight"), 4, 0.25, "sym", 2.0)
    assert op[2][-4:].tolist() == [-1.0] * 4
    o_r, o_i = R.magnet_conv(g.t("x_real"), g.t("x_imag"), op, g.t("weight"), g.t("bias"))
    np.testing.assert_allclose(o_r.numpy(), [[11.624232, -1.320588], [9.814465, 0.440558],
                                             [11.364678, 7.492043], [7.1, 6.8]], atol=2e-6)
    np.testing.assert_allclose(o_i.numpy(), [[11.754375, -0.191489], [10.056267, 3.859610],
                                             [14.364678, 1.492042], [7.1, 8.8]], atol=2e-6)
    _, sre, sim = R.magnetic_laplacian(g.t("edge_index"), g.t("signed_weight"), 4, 0.25, "sym", True)
    np.testing.assert_allclose(sre.numpy(), [1.128623e-08, 0.3651484, 1.128623e-08, -0.4714045,
                                             0.3651484, -0.4714045, 1, 1, 1, 1], rtol=2e-6, atol=1e-12)
    assert_close(sim, g["signed_lap_imag"])


@pytest.mark.parametrize("name", golden_names("digcn_"))
def test_digcn(name):
    g = load_golden(name)
    x, w = g.t("x").requires_grad_(), g.t("weight").requires_grad_()
    b = g.t("bias")
    out = R.digcn_conv(x, g.t("edge_index"), g.t("edge_weight"), w, b)
    assert_close(out, g["out"])
    (out * g.t("grad_out")).sum().backward()
    assert_close(x.grad, g["dx"])
    assert_close(w.grad, g["dweight"], 2e-6)


@pytest.mark.parametrize("name", golden_names("dgcn_"))
def test_dgcn(name):
    g = load_golden(name)
    x = g.t("x").requires_grad_()
    out = R.dgcn_conv(x, g.t("edge_index"), g.t("edge_weight"), bool(g["improved"]),
                      bool(g["add_self_loops"]))
    assert_close(out, g["out"])
    (out * g.t("grad_out")).sum().backward()
    assert_close(x.grad, g["dx"])


@pytest.mark.parametrize("name", golden_names("conv_base_"))
def test_conv_base(name):
    g = load_golden(name)
    x = g.t("x").requires_grad_()
    out = R.conv_base(x, g.t("edge_index"), g.t("edge_weight"), float(g["fill_value"]))
    assert_close(out, g["out"])
    (out * g.t("grad_out")).sum().backward()
    assert_close(x.grad, g["dx"])


@pytest.mark.parametrize("name", golden_names("simpa_"))
def test_simpa(name):
    g = load_golden(name)
    directed = bool(g["directed"])
    params = {k[5:]: g.t(k).requires_grad_() for k in g if k.startswith("param")}
    xs = [g.t(k).requires_grad_() if k in g else None for k in ("x_p", "x_n", "x_pt", "x_nt")]
    out = R.simpa(g.t("edge_index_p"), g.t("edge_weight_p"), g.t("edge_index_n"),
                  g.t("edge_weight_n"), xs[0], xs[1], params, int(g["hop"]),
                  float(g["fill_value"]), directed, xs[2], xs[3])
    assert_close(out, g["out"])
    (out * g.t("grad_out")).sum().backward()
    assert_close(xs[0].grad, g["dx_p"])
    assert_close(xs[1].grad, g["dx_n"])
    for k, p in params.items():
        assert_close(p.grad, g["dparam" + k], 3e-6)


def test_dimpa():
    g = load_golden("dimpa_hop2")
    xs, xt = g.t("x_s").requires_grad_(), g.t("x_t").requires_grad_()
    out = R.dimpa(xs, xt, g.t("edge_index"), g.t("edge_weight"), g.t("w_s"), g.t("w_t"),
                  int(g["hop"]), float(g["fill_value"]))
    assert_close(out, g["out"])
    (out * g.t("grad_out")).sum().backward()
    assert_close(xs.grad, g["dx_s"])
    assert_close(xt.grad, g["dx_t"])


@pytest.mark.parametrize("name", golden_names("sgcn_"))
def test_sgcn(name):
    g = load_golden(name)
    x = g.t("x").requires_grad_()
    out = R.sgcn_conv(x, g.t("pos_edge_index"), g.t("neg_edge_index"),
                      (g.t("lin_b_weight"), g.t("lin_b_bias")), (g.t("lin_u_weight"), g.t("lin_u_bias")),
                      bool(g["first_aggr"]), int(g["in_dim"]), bool(g["norm_emb"]))
    assert_close(out, g["out"])
    (out * g.t("grad_out")).sum().backward()
    assert_close(x.grad, g["dx"])


def test_complex_relu():
    g = load_golden("complex_relu")
    o_r, o_i = R.complex_relu(g.t("real"), g.t("imag"))
    assert np.array_equal(o_r.numpy(), g["out_real"]) and np.array_equal(o_i.numpy(), g["out_imag"])


def test_gat_conv_and_sdr_layer():
    """Attention aggregate (reference SDRLayer over the restated PyG GATConv) vs the oracle."""
    g = load_golden("gat_conv")
    x = g.t("x").requires_grad_()
    prm = {k[3:]: g.t(k).requires_grad_() for k in g if k.startswith("sd.")}
    out = R.gat_conv(x, g.t("edge_index"), prm["lin.weight"], prm["att_src"], prm["att_dst"], prm["bias"])
    assert_close(out, g["out"])
    (out * g.t("grad_out")).sum().backward()
    assert_close(x.grad, g["dx"])
    for k, p in prm.items():
        assert_close(p.grad, g["d." + k], 2e-6)
    g = load_golden("sdr_layer")
    x = g.t("x")
    neigh = [R.gat_conv(x, g.t(f"edges{k}"), g.t(f"sd.agg_{k}.lin.weight"), g.t(f"sd.agg_{k}.att_src"),
                        g.t(f"sd.agg_{k}.att_dst"), g.t(f"sd.agg_{k}.bias")) for k in range(4)]
    hcat = torch.cat([x] + neigh, 1)
    hid = torch.tanh(torch.nn.functional.linear(hcat, g.t("sd.mlp_layer.0.weight"), g.t("sd.mlp_layer.0.bias")))
    out = torch.nn.functional.linear(hid, g.t("sd.mlp_layer.2.weight"), g.t("sd.mlp_layer.2.bias"))
    assert_close(out, g["out"])
