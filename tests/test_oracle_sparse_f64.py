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
