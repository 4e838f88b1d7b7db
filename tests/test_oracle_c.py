"""CPU: the C restatement (oracle/pygsd_oracle.c) against the golden vectors recorded from the
reference and against the torch restatement (oracle/ref_layers.py)."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from oracle import c_oracle as C
from oracle import ref_layers as R


def test_c_propagate_bitwise_vs_torch_restatement():
    g = torch.Generator().manual_seed(0)
    n, e, f = 60, 700, 9
    ei = torch.randint(0, n, (2, e), generator=g)
    x, w = torch.randn(n, f, generator=g), torch.randn(e, generator=g)
    for flow in ("source_to_target", "target_to_source"):
        got = C.propagate(x.numpy(), ei.numpy(), w.numpy(), n, flow)
        want = R.propagate(x, ei, w, n, flow=flow).numpy()
        assert np.array_equal(got, want)  # same op order -> bit-identical
        got = C.propagate(x.numpy(), ei.numpy(), None, n, flow, mean=True)
        want = R.propagate(x, ei, None, n, flow=flow, reduce="mean").numpy()
        assert np.array_equal(got, want)


@pytest.mark.parametrize("name", golden_names("magnet_") + golden_names("msconv_"))
def test_c_laplacian_vs_reference_operator(name):
    g = load_golden(name)
    if str(g["normalization"]) == "none":
        lam = float(g["lambda_max"])
    else:
        lam = 2.0
    ei, re, im = C.magnetic_laplacian(g["edge_index"], g.get("edge_weight"), 40, float(g["q"]),
                                      None if str(g["normalization"]) == "none" else "sym",
                                      bool(g["signed"]), bool(g["absolute_degree"]))
    assert ei.tolist() == g["op_index_imag"].tolist()
    lam32 = np.float32(lam)
    np.testing.assert_allclose((np.float32(2.0) * re) / lam32, g["op_real"][:re.size], rtol=0, atol=1e-6)
    np.testing.assert_allclose((np.float32(2.0) * im) / lam32, g["op_imag"], rtol=0, atol=1e-6)


def test_c_kat():
    g = load_golden("kat_appendix_b")
    ei, re, im = C.magnetic_laplacian(g["edge_index"], g["edge_weight"], 4, 0.25)
    assert ei.tolist() == g["lap_index"].tolist()
    np.testing.assert_allclose(re, g["lap_real"], rtol=2e-6, atol=1e-12)
    np.testing.assert_allclose(im, g["lap_imag"], rtol=2e-6, atol=1e-12)


def test_c_complex_relu():
    g = load_golden("complex_relu")
    o_r, o_i = C.complex_relu(g["real"], g["imag"])
    assert np.array_equal(o_r, g["out_real"]) and np.array_equal(o_i, g["out_imag"])
