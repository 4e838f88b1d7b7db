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
