"""CPU: host-side contract of the drop-in layers -- constructor signatures, __repr__ strings,
state_dict keys / shapes / initialisation, error behaviour, and that the product
path refuses CPU tensors instead of silently falling back."""
import math

import pytest
import torch

from pytorch_geometric_signed_directed_amd.nn import (DIMPA, SIMPA, Conv_Base, DGCNConv, DiGCNConv, MagNetConv,
                                                      MSConv, SGCNConv, complex_relu_layer)


def test_repr_strings_asserted_by_the_reference_tests():
    # reference test/directed_test.py:73-74, :281; test/signed_test.py (SGCNConv repr)
    assert repr(MagNetConv(3, 2, 2, 0.25, False)) == 'MagNetConv(3, 2, filter size=3, normalization=sym)'
    assert repr(MSConv(3, 2, 2, 0.25, False)) == 'MSConv(3, 2, filter size=3, normalization=sym)'
    assert repr(DiGCNConv(3, 4)) == 'DiGCNConv(3, 4)'
    assert repr(SGCNConv(4, 3, first_aggr=True)) == 'SGCNConv(4, 3, first_aggr=True)'


def test_state_dict_layout_and_init():
    m = MagNetConv(7, 5, 3, 0.1, True)
    sd = m.state_dict()
    assert list(sd) == ['q', 'weight', 'bias']
    assert sd['weight'].shape == (4, 7, 5) and sd['bias'].shape == (5,) and sd['q'].shape == (1,)
    a = math.sqrt(6.0 / (7 + 5))
    assert float(sd['weight'].abs().max()) <= a and float(sd['bias'].abs().max()) == 0.0
    assert list(MagNetConv(7, 5, 1, 0.1, False, bias=False).state_dict()) == ['weight']
    assert list(MSConv(7, 5, 1, 0.1, False).state_dict()) == ['weight', 'bias']
    d = DiGCNConv(6, 4)
    assert {k: tuple(v.shape) for k, v in d.state_dict().items()} == {'weight': (6, 4), 'bias': (4,)}
    s = SGCNConv(4, 3, first_aggr=False)
    assert {k: tuple(v.shape) for k, v in s.state_dict().items()} == {
        'lin_b.weight': (3, 12), 'lin_b.bias': (3,), 'lin_u.weight': (3, 12), 'lin_u.bias': (3,)}
    assert SGCNConv(4, 3, first_aggr=True).lin_b.weight.shape == (3, 8)
    u = SIMPA(2, 0.5)
    assert {k: tuple(v.shape) for k, v in u.state_dict().items()} == {'_w_p': (3, 1), '_w_n': (3, 1)}
    assert all(float(v.min()) == 1.0 == float(v.max()) for v in u.state_dict().values())
    assert list(SIMPA(2, 0.5, directed=True).state_dict()) == ['_w_sp', '_w_sn', '_w_tp', '_w_tn']
    assert {k: tuple(v.shape) for k, v in DIMPA(3).state_dict().items()} == {'_w_s': (4, 1), '_w_t': (4, 1)}
    assert len(DGCNConv().state_dict()) == 0 and len(Conv_Base().state_dict()) == 0


def test_constructor_defaults_match_the_reference():
    m = MagNetConv(3, 2, 1, 0.25, False)
    assert (m.normalization, m.cached, m.trainable_q, m.flow, m.aggr) == ('sym', False, False,
                                                                          'source_to_target', 'add')
    ms = MSConv(3, 2, 1, 0.25, False)
    assert ms.absolute_degree is True and ms.cached is False
    assert DiGCNConv(3, 2).cached is True and DiGCNConv(3, 2).improved is False
    d = DGCNConv()
    assert (d.improved, d.cached, d.add_self_loops, d.normalize) == (False, False, True, True)
    c = Conv_Base()
    assert (c.fill_value, c.flow, c.aggr) == (0.5, 'target_to_source', 'add')
    assert SGCNConv(4, 3, True).aggr == 'mean'
    with pytest.raises(AssertionError):
        MagNetConv(3, 2, 0, 0.25, False)
    with pytest.raises(AssertionError):
        MagNetConv(3, 2, 1, 0.25, False, normalization='rw')


def test_reference_checkpoint_roundtrip():
    a, b = MagNetConv(4, 4, 2, 0.25, False), MagNetConv(4, 4, 2, 0.25, False)
    b.load_state_dict(a.state_dict())
    assert torch.equal(a.weight, b.weight)


def test_cpu_tensors_are_refused_loudly():
    x = torch.randn(5, 3)
    ei = torch.tensor([[0, 1, 2], [1, 2, 3]])
    w = torch.ones(3)
    for call in (lambda: MagNetConv(3, 2, 1, 0.25, False)(x, x, ei, w),
                 lambda: MSConv(3, 2, 1, 0.25, False)(x, x, ei, w),
                 lambda: DiGCNConv(3, 2)(x, ei, w),
                 lambda: DGCNConv()(x, ei, w),
                 lambda: Conv_Base()(x, ei, w),
                 lambda: SGCNConv(3, 2, True)(x, ei, ei),
                 lambda: complex_relu_layer()(x, x)):
        with pytest.raises(RuntimeError, match="no CPU fallback|HIP"):
            call()


def test_product_package_does_not_import_the_oracle():
    import subprocess
    import sys
    code = ("import sys, pytorch_geometric_signed_directed_amd as p, pytorch_geometric_signed_directed_amd.nn;"
            "bad=[m for m in sys.modules if m=='oracle' or m.startswith('oracle.')];"
            "assert not bad, bad")
    subprocess.run([sys.executable, "-c", code], check=True)


def test_negative_samplers_and_spectral_features():
    """Host-side helpers of the signed callers (CPU tensors are fine here: no device kernel involved)."""
    import numpy as np
    import torch
    from conftest import load_golden
    from pytorch_geometric_signed_directed_amd.utils.signed import (create_spectral_features, negative_sampling,
                                                                     structured_negative_sampling)
    g = torch.Generator().manual_seed(0)
    n = 30
    ei = torch.randint(0, n, (2, 200), generator=g)
    listed = set((ei[0] * n + ei[1]).tolist())
    ns = negative_sampling(ei, n, generator=g)
    assert ns.shape == (2, 200) and not (set((ns[0] * n + ns[1]).tolist()) & listed)
    assert len(set((ns[0] * n + ns[1]).tolist())) == 200
    i, j, k = structured_negative_sampling(ei, n, generator=g)
    assert torch.equal(i, ei[0]) and torch.equal(j, ei[1]) and not (set((i * n + k).tolist()) & listed)
    # spectral features: singular vectors are defined up to sign
    gold = load_golden("model_sgcn")
    es = gold.t("edge_index_s")
    pos, neg = es[es[:, 2] > 0][:, :2].t(), es[es[:, 2] < 0][:, :2].t()
    got = create_spectral_features(pos, neg, 40, 5).numpy()
    want = gold["spectral"]
    for c in range(5):
        assert min(np.abs(got[:, c] - want[:, c]).max(), np.abs(got[:, c] + want[:, c]).max()) < 1e-3


def test_sdgnn_motif_weights_match_recorded_reference_matrix():
    """Host-side construction only (no device work): neighbour edge lists as sets, triangle-motif counts."""
    import scipy.sparse as sp
    import torch
    from conftest import load_golden
    from pytorch_geometric_signed_directed_amd.nn.models import SDGNN
    g = load_golden("model_sdgnn")
    es = g.t("edge_index_s")
    m = SDGNN.__new__(SDGNN)
    torch.nn.Module.__init__(m)
    m.node_num, m.device = 40, torch.device("cpu")
    lists = m.build_edge_lists(es)
    want = sp.coo_matrix((g["tri_val"], (g["tri_row"], g["tri_col"])), shape=(40, 40)).tocsr()
    assert abs(m.tri_weight.tocsr() - want).sum() == 0
    pos = {(int(a), int(b)) for a, b, s in es.tolist() if s > 0}
    neg = {(int(a), int(b)) for a, b, s in es.tolist() if s < 0}
    as_set = lambda t: set(map(tuple, t.t().tolist()))  # noqa: E731
    assert as_set(lists[0]) == pos and as_set(lists[1]) == {(b, a) for a, b in pos}
    assert as_set(lists[2]) == neg and as_set(lists[3]) == {(b, a) for a, b in neg}
    assert all(t.size(1) == len(as_set(t)) for t in lists)          # duplicate listings collapsed


def test_digcn_operator_preprocessing_matches_reference():
    """get_second_directed_adj / get_appr_directed_adj / cal_fast_appr (host-side graph preparation): same
    entry layout as the reference's dense code (torch.nonzero order) and values within 5e-6; the sparse power
    iteration used beyond 2000 nodes finds the dense solver's Perron vector."""
    import numpy as np
    import torch
    from conftest import load_golden
    from pytorch_geometric_signed_directed_amd.utils.directed import get_adjs_DiGCN as A
    g = load_golden("adjs_digcn")
    ei, w = g.t("edge_index"), g.t("edge_weight")
    cases = {"second": A.get_second_directed_adj(ei, 40, torch.float32, w),
             "second_unw": A.get_second_directed_adj(ei, 40, torch.float32, None),
             "appr": A.get_appr_directed_adj(0.1, ei, 40, torch.float32, w),
             "appr_unw": A.get_appr_directed_adj(0.2, ei, 40, torch.float32, None),
             "fast": A.cal_fast_appr(0.1, ei, 40, torch.float32, w)}
    for name, (index, value) in cases.items():
        assert np.array_equal(index.numpy(), g[name + "_index"]), name
        assert np.abs(value.numpy() - g[name + "_value"]).max() < 5e-6, name     # the reference rounds in float32
    p = A._transition(ei, w, 40, torch.float32)
    dense = A._perron_left_vector(p, 0.1, 40)
    power = A._perron_left_vector(p, 0.1, 40, dense_limit=0)             # force the sparse power iteration
    assert np.abs(power / power.sum() - dense / dense.sum()).max() < 1e-6


def test_tall_product_and_column_sums_library_route_on_cpu():
    """dense.tall_product / column_sums off the GPU are plain torch (the HIP kernels take CUDA tensors only): segments are
    multiplied block by block and accumulated, W^T for the input gradient, split outputs are column views."""
    import torch
    from pytorch_geometric_signed_directed_amd.dense import column_sums, column_sums_of, tall_product
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(50, 32, generator=g), torch.randn(50, 64, generator=g)
    w, bias = torch.randn(96, 48, generator=g), torch.randn(48, generator=g)
    want = torch.cat([a, b], 1).double() @ w.double() + bias.double()
    assert torch.allclose(tall_product([a, b], w, False, bias).double(), want, atol=1e-4)
    assert torch.allclose(tall_product([a, b], w.t().contiguous(), True, bias).double(), want, atol=1e-4)
    parts = tall_product([a, b], w, False, bias, splits=(16, 32))
    assert [tuple(p.shape) for p in parts] == [(50, 16), (50, 32)]
    assert torch.allclose(torch.cat(parts, 1).double(), want, atol=1e-4)
    import pytest
    with pytest.raises(ValueError):
        tall_product([a, b], w[:90], False)
    with pytest.raises(ValueError):
        tall_product([a, b], w, False, None, splits=(16, 16))
    assert torch.equal(column_sums(a), a.sum(0))
    x, y, z = column_sums_of([a, None, a])
    assert y is None and x is z


def test_library_routes_are_counted_and_announced_once():
    """_cabi.note_library_route: every library-routed dense product is counted; the warning fires once per (site, shape)."""
    import warnings
    from pytorch_geometric_signed_directed_amd import _cabi
    _cabi.reset_library_routes()
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        _cabi.note_library_route("unit-test site", "(3, 5) float32")
        _cabi.note_library_route("unit-test site", "(3, 5) float32")
        _cabi.note_library_route("unit-test site", "(4, 5) float32")
    assert _cabi.library_routes() == {"unit-test site": 3}
    assert len([w for w in seen if "unit-test site" in str(w.message)]) == 2
    _cabi.reset_library_routes()
    assert _cabi.library_routes() == {}


def test_c16_gives_contiguous_aligned_canonically_strided_operands():
    """`_cabi.c16`: what the float4 kernels are handed.  A one-row column slice is contiguous wherever it starts and keeps its
    parent's row stride -- both have to go (tests/test_gpu_layers.py::test_one_row_column_slices_reach_the_kernels_aligned runs
    the layers on such inputs); tensors that already qualify are passed through untouched."""
    import torch
    from pytorch_geometric_signed_directed_amd import _cabi
    wide = torch.arange(2 * 35, dtype=torch.float32).reshape(2, 35)
    assert _cabi.c16(None) is None
    whole = torch.zeros(4, 16)
    assert _cabi.c16(whole) is whole
    one = wide[:1, 1:17]                                   # one row, 4 bytes off the buffer's start, row stride 35
    assert one.is_contiguous() and one.stride(0) == 35
    got = _cabi.c16(one)
    assert got.data_ptr() % 16 == 0 and got.stride() == (16, 1) and torch.equal(got, one) and got is not one
    two = wide[:, 4:20]                                    # two rows: not contiguous
    got = _cabi.c16(two)
    assert got.is_contiguous() and got.data_ptr() % 16 == 0 and torch.equal(got, two)
    row = torch.zeros(1, 16)
    assert _cabi.c16(row) is row                           # a fresh one-row matrix has canonical strides
