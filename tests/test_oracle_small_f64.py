"""CPU: oracle/small_f64_torch.py (the float64 arbiter of the toy-sized GPU checks whose fixtures hold fp32 reference
results only) pinned against the dense numpy formulas and the fixtures recorded from the reference."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import load_golden
from oracle import dense_f64 as D64
from oracle import small_f64_torch as F64


def _near(got, want, tol=2e-5):
    want = np.asarray(want, np.float64)
    got = got.detach().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    assert np.abs(got.reshape(want.shape) - want).max() <= tol * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("name", ["snea_first", "snea_deep"])
def test_snea_float64_with_gradients(name):
    g = load_golden(name)
    first = bool(g["first_aggr"])
    prm = {k[3:]: g.t(k).double().requires_grad_() for k in g if k.startswith("sd.")}
    x = g.t("x").double().requires_grad_()
    out = F64.snea_conv(x, g["pos"], g["neg"], (prm["lin_b.weight"], prm["lin_b.bias"]), (prm["lin_u.weight"], prm["lin_u.bias"]),
                        (prm["alpha_b.weight"], prm["alpha_b.bias"]), (prm["alpha_u.weight"], prm["alpha_u.bias"]), first, 5)
    assert np.abs(out.detach().numpy() - g["dense_f64"]).max() <= 1e-12       # the numpy node-by-node evaluation
    out.backward(g.t("gout").double())
    _near(x.grad, g["dx"])
    for k, p in prm.items():
        _near(p.grad, g["grad." + k])


def test_cut_losses_float64_with_gradients():
    g = load_golden("sssnet_losses")
    ei, w = g["edge_index"], g["edge_weight"]
    a = sp.coo_matrix((w, (ei[0], ei[1])), shape=(40, 40)).toarray().astype(np.float64)
    a_p, a_n = np.maximum(a, 0), np.maximum(-a, 0)
    for j, name in enumerate(("normalized", "ratio", "unhappy")):
        prob = g.t("prob").double().requires_grad_()
        val = F64.cut_losses(a_p, a_n, prob)[j]
        _near(val, g["loss_" + name])
        val.backward()
        _near(prob.grad, g["dprob_" + name])


def test_imbalance_loss_float64_with_gradients():
    g = load_golden("digrac_imbalance_loss")
    ei, w = g["edge_index"], g["edge_weight"]
    a = np.zeros((40, 40))
    np.add.at(a, (ei[0], ei[1]), w)
    for norm in ("vol_sum", "vol_min", "vol_max", "plain"):
        for thr in ("sort", "std", "naive"):
            prob = g.t(f"prob_{norm}_{thr}").double().requires_grad_()
            val = F64.imbalance_loss(prob, a, 4, 3, norm, thr)
            _near(val, g[f"loss_{norm}_{thr}"])
            if thr == "sort":
                val.sum().backward()
                _near(prob.grad, g[f"dprob_{norm}_{thr}"], 5e-5)
