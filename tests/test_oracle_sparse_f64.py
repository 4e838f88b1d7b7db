"""CPU: oracle/sparse_f64.py (the float64 scipy evaluation used by the full-size GPU parity tests) is held to
oracle/dense_f64.py (independent dense formulas) and to the golden fixtures recorded from the reference."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle import dense_f64 as D64
from oracle import sparse_f64 as S64


@pytest.mark.parametrize("signed,norm,absdeg", [(False, "sym", True), (False, None, True), (True, "sym", True),
                                                (True, "sym", False), (True, None, False)])
def test_sparse_operator_and_layer_equal_dense_formulas(signed, norm, absdeg):
    rng = np.random.default_rng(5)
    n, e, f = 70, 600, 5
    ei = rng.integers(0, n, (2, e))
    w = rng.uniform(0.5, 1.5, e) * (rng.choice([-1, 1], e) if signed else 1)
    lam = 2.0 if norm == "sym" else 6.5
    dense = D64.magnetic_operator(ei, w, n, 0.2, norm, lam, signed, absdeg)
    sparse = S64.magnetic_operator(ei, w, n, 0.2, norm, lam, signed, absdeg)
    assert np.abs(sparse.toarray() - dense).max() <= 1e-12
    for k in (1, 3):
        xr, xi = rng.normal(size=(n, f)), rng.normal(size=(n, f))
        wt, b = rng.normal(size=(k + 1, f, 4)), rng.normal(size=4)
        got = S64.magnet_conv(xr, xi, sparse, wt, b)
        want = D64.magnet_conv(xr, xi, dense, wt, b)
        assert max(np.abs(got[0] - want[0]).max(), np.abs(got[1] - want[1]).max()) <= 1e-10


def test_sampled_build_equals_full_build_on_the_sampled_rows():
    rng = np.random.default_rng(6)
    n, e, f = 2000, 30000, 8
    ei = rng.integers(0, n, (2, e))
    w = rng.uniform(0.5, 1.5, e)
    rows = rng.choice(n, 40, replace=False)
    full = S64.magnetic_operator(ei, w, n, 0.25)
    part = S64.magnetic_operator(ei, w, n, 0.25, only_nodes=rows)
    xr, xi, gr, gi = (rng.normal(size=(n, f)) for _ in range(4))
    wt, b = rng.normal(size=(2, f, f)), rng.normal(size=f)
    everything = S64.magnet_conv(xr, xi, full, wt, b, gr, gi)
    sampled = S64.magnet_conv_rows_k1(xr, xi, part, wt, b, rows, gr, gi)
    for k in range(4):
        assert np.abs(sampled[k] - everything[k][rows]).max() <= 1e-12


@pytest.mark.parametrize("name", [n for n in golden_names("magnet_") + golden_names("msconv_") if "wide" not in n])
def test_sparse_f64_reproduces_the_reference_fixtures(name):
    """Outputs AND gradients recorded from the reference's own Python (fp32) within 2e-5 of the float64 result."""
    g = load_golden(name)
    if "lambda_max" in g and str(g["normalization"]) == "none":
        lam = float(g["lambda_max"])
    else:
        lam = 2.0
    norm = None if str(g["normalization"]) == "none" else "sym"
    n = g["x_real"].shape[0]
    s = S64.magnetic_operator(g["edge_index"], g.get("edge_weight"), n, float(g["q"]), norm, lam,
                              bool(g["signed"]), bool(g["absolute_degree"]))
    got = S64.magnet_conv(g["x_real"], g["x_imag"], s, g["weight"], g.get("bias"), g["grad_real"], g["grad_imag"])
    names = ["out_real", "out_imag", "dx_real", "dx_imag", "dweight"] + (["dbias"] if "dbias" in g else [])
    for arr, key in zip(got, names):
        want = np.asarray(g[key], np.float64)
        assert np.abs(arr - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), key


@pytest.mark.parametrize("signed,norm,absdeg", [(False, "sym", True), (False, None, True), (True, "sym", True),
                                                (True, "sym", False), (True, None, False)])
def test_torch_restatement_equals_the_scipy_evaluation(signed, norm, absdeg):
    """oracle/sparse_f64_torch.py (the float64 evaluation that also runs on the device, for the checks at the BASELINE
    configs' stated sizes) against oracle/sparse_f64.py: operator, outputs and every gradient, K = 1 and 3; its
    DiGCN product against the reference op sequence of ref_layers.py in float64."""
    import torch
    from oracle import ref_layers as R
    from oracle import sparse_f64_torch as T64
    rng = np.random.default_rng(8)
    n, e, f = 90, 900, 6
    ei = rng.integers(0, n, (2, e))
    ei[:, :40] = ei[::-1, 40:80]                                 # reciprocal pairs
    ei[:, 80:100] = ei[:, 100:120]                               # exact duplicates
    ei[1, 120:130] = ei[0, 120:130]                              # self loops
    w = rng.uniform(0.5, 1.5, e) * (rng.choice([-1, 1], e) if signed else 1)
    lam = 2.0 if norm == "sym" else 5.5
    s = S64.magnetic_operator(ei, w, n, 0.2, norm, lam, signed, absdeg)
    t = T64.magnetic_operator(torch.from_numpy(ei), torch.from_numpy(w), n, 0.2, norm, lam, signed, absdeg)
    dense = np.zeros((n, n), np.complex128)
    np.add.at(dense, (t.row.numpy(), t.col.numpy()), t.real.numpy() + 1j * t.imag.numpy())
    dense[np.arange(n), np.arange(n)] += t.diag.numpy()
    assert np.abs(dense - s.toarray()).max() <= 1e-12
    for k in (1, 3):
        xr, xi, gr, gi = (rng.normal(size=(n, f)) for _ in range(4))
        wt, b = rng.normal(size=(k + 1, f, f)), rng.normal(size=f)
        want = S64.magnet_conv(xr, xi, s, wt, b, gr, gi)
        got = T64.magnet_conv(*(torch.from_numpy(a) for a in (xr, xi)), t, torch.from_numpy(wt), torch.from_numpy(b),
                              torch.from_numpy(gr), torch.from_numpy(gi))
        for a, c in zip(got, want):
            assert np.abs(a.numpy() - c).max() <= 1e-11
    x = torch.from_numpy(rng.normal(size=(n, f))).requires_grad_()
    wt = torch.from_numpy(rng.normal(size=(f, 4))).requires_grad_()
    bias = torch.from_numpy(rng.normal(size=4)).requires_grad_()
    go = torch.from_numpy(rng.normal(size=(n, 4)))
    eit, ew = torch.from_numpy(ei), torch.from_numpy(np.abs(w))
    want = R.digcn_conv(x, eit, ew, wt, bias)
    (want * go).sum().backward()
    got = T64.digcn_conv(x.detach(), eit, ew, wt.detach(), bias.detach(), go)
    for a, c in zip(got, (want.detach(), x.grad, wt.grad, bias.grad)):
        assert (a - c).abs().max() <= 1e-11


def _signed_graph(rng, n, e_pos, e_neg):
    """Positive / negative edge lists with reciprocal pairs, exact duplicates and listed self loops (twice on one node)."""
    def part(e):
        ei = rng.integers(0, n, (2, e))
        ei[:, :10] = ei[::-1, 10:20]
        ei[:, 20:26] = ei[:, 26:32]
        ei[1, 32:38] = ei[0, 32:38]
        ei[:, 38] = ei[:, 37]                     # the same loop listed twice: the LAST weight wins
        return ei, rng.uniform(0.5, 1.5, e)
    return part(e_pos), part(e_neg)


@pytest.mark.parametrize("directed,hop", [(False, 2), (False, 3), (True, 2)])
def test_torch_signed_layers_equal_the_dense_formulas_and_the_reference_sequence(directed, hop):
    """oracle/sparse_f64_torch.py's SIMPA / DIMPA / SGCNConv / SSSNET (the float64 arbiter of the C3 checks at the stated
    size): forward against the independent dense formulas of oracle/dense_f64.py, gradients against autograd through
    the reference op sequence (oracle/ref_layers.py) evaluated in float64."""
    import torch
    from oracle import ref_layers as R
    from oracle import sparse_f64_torch as T64
    rng = np.random.default_rng(12)
    n, f = 70, 5
    (ei_p, w_p), (ei_n, w_n) = _signed_graph(rng, n, 400, 250)
    tp, tn = torch.from_numpy(ei_p), torch.from_numpy(ei_n)
    twp, twn = torch.from_numpy(w_p), torch.from_numpy(w_n)
    names = ("_w_sp", "_w_sn", "_w_tp", "_w_tn") if directed else ("_w_p", "_w_n")
    rows = {True: hop + 1, False: (hop + 1) * hop // 2}
    xs_np = [rng.normal(size=(n, f)) for _ in range(4 if directed else 2)]
    prm_np = {k: rng.uniform(0.5, 1.5, (rows[k.endswith("p")], 1)) for k in names}
    go = torch.from_numpy(rng.normal(size=(n, f * len(xs_np))))
    want = D64.simpa(ei_p, w_p, ei_n, w_n, xs_np[0], xs_np[1], prm_np, hop, 0.5, directed, *xs_np[2:])

    def run(mod):
        xs = [torch.from_numpy(a).requires_grad_() for a in xs_np]
        prm = {k: torch.from_numpy(v).requires_grad_() for k, v in prm_np.items()}
        out = mod.simpa(tp, twp, tn, twn, xs[0], xs[1], prm, hop, 0.5, directed, *xs[2:])
        (out * go).sum().backward()
        return [out.detach()] + [x.grad for x in xs] + [prm[k].grad for k in names]

    got, ref = run(T64), run(R)
    assert np.abs(got[0].numpy() - want).max() <= 1e-11
    for a, c in zip(got, ref):
        assert (a - c).abs().max() <= 1e-11
    if directed:
        return
    # DIMPA
    x_s, x_t = rng.normal(size=(n, f)), rng.normal(size=(n, f))
    w_s, w_t = rng.uniform(0.5, 1.5, (hop + 1, 1)), rng.uniform(0.5, 1.5, (hop + 1, 1))
    want = D64.dimpa(x_s, x_t, ei_p, w_p, w_s, w_t, hop, 0.5)
    god = torch.from_numpy(rng.normal(size=(n, 2 * f)))

    def run_d(fn):
        a, b = torch.from_numpy(x_s).requires_grad_(), torch.from_numpy(x_t).requires_grad_()
        ws, wt = torch.from_numpy(w_s).requires_grad_(), torch.from_numpy(w_t).requires_grad_()
        out = fn(a, b, tp, twp, ws, wt, hop, 0.5)
        (out * god).sum().backward()
        return out.detach(), a.grad, b.grad, ws.grad, wt.grad

    got, ref = run_d(T64.dimpa), run_d(R.dimpa)
    assert np.abs(got[0].numpy() - want).max() <= 1e-11
    for a, c in zip(got, ref):
        assert (a - c).abs().max() <= 1e-11
    # SGCNConv, first and deep aggregation
    for first in (True, False):
        in_dim, o = 6, 4
        x_np = rng.normal(size=(n, in_dim if first else 2 * in_dim))
        k = 2 if first else 3
        lb = (rng.normal(size=(o, k * in_dim)), rng.normal(size=o))
        lu = (rng.normal(size=(o, k * in_dim)), rng.normal(size=o))
        want = D64.sgcn_conv(x_np, ei_p, ei_n, lb, lu, first, in_dim)
        gos = torch.from_numpy(rng.normal(size=(n, 2 * o)))

        def run_s(fn):
            x = torch.from_numpy(x_np).requires_grad_()
            prm = [torch.from_numpy(a).requires_grad_() for a in lb + lu]
            out = fn(x, tp, tn, (prm[0], prm[1]), (prm[2], prm[3]), first, in_dim)
            (out * gos).sum().backward()
            return [out.detach(), x.grad] + [p.grad for p in prm]

        got, ref = run_s(T64.sgcn_conv), run_s(R.sgcn_conv)
        assert np.abs(got[0].numpy() - want).max() <= 1e-11
        for a, c in zip(got, ref):
            assert (a - c).abs().max() <= 1e-11


@pytest.mark.parametrize("name", ["model_sssnet_undirected", "model_sssnet_directed"])
def test_torch_sssnet_reproduces_the_reference_model_fixtures(name):
    """The float64 SSSNET restatement against the outputs recorded from the reference's own model (fp32, eval mode)."""
    import torch
    from oracle import sparse_f64_torch as T64
    g = load_golden(name)
    sd = {k[3:]: torch.from_numpy(np.asarray(g[k], np.float64)) for k in g if k.startswith("sd.")}
    directed = bool(g["directed"])
    hop = sd["_simpa._w_sp" if directed else "_simpa._w_p"].numel() - 1
    fill = float(g["fill_value"]) if "fill_value" in g else 0.5
    z, logp, prob = T64.sssnet(g.t("edge_index_p"), g.t("edge_weight_p"), g.t("edge_index_n"), g.t("edge_weight_n"),
                               g.t("x").double(), sd, hop, fill, directed)
    for got, key in ((z, "out0"), (logp, "out1"), (prob, "out3")):
        want = np.asarray(g[key], np.float64)
        assert np.abs(got.numpy() - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), key
