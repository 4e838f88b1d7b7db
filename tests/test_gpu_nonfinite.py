"""GPU: non-finite and extreme-range operands through the DEFAULT (split) forms of the fp32 dense products.

The default fp32 form of pygsd_tall_linear / pygsd_tall_gram / pygsd_magnetic_dense_{fwd,bwd}_f32 multiplies on the bf16 matrix
pipe (every fp32 value = three bf16 pieces, csrc/tall.hip).  The pieces carry FINITE values below the largest bf16 only; what
the reference computes with `torch.matmul` (nn/directed/MagNetConv.py:217-247, nn/directed/DiGCNConv.py:66,
nn/signed/SGCNConv.py:121-126 and their autograd) carries +-inf, NaN and magnitudes up to FLT_MAX by IEEE rules.  These tests
plant such values and hold the kernels to the same fp32 `torch.matmul` on the CPU:

  * the SAME pattern of NaN, +inf and -inf, element for element;
  * every other element within the suite's bar, |d| <= 1e-5 (1 + |want|)  (max-norm for reductions over the rows).

The planted weights include values that fit 8 mantissa bits (their mid / lo pieces are exactly zero: inf x piece would be NaN
where inf x w is +-inf) and an exact zero (inf x 0 = NaN in the reference as well)."""
import pytest
import torch

from tolerance import single_thread

pytestmark = pytest.mark.gpu
TOL = 1e-5
INF, NAN = float("inf"), float("nan")
BIG = 3.3e38               # below the largest bf16 (3.3895e38): the split carries it
HUGE = 3.4e38              # above it, below FLT_MAX: the hi piece rounds to inf
TINY = 1e-38               # a normal number whose lower pieces are subnormal
SUB = 1e-40                # subnormal


def dev():
    return torch.device("cuda:0")


def same_special_values_and_close(got, want, what, norm=False, truth64=None):
    """The fp32 reference decides where NaN / +inf / -inf stand.  The finite elements are held to the suite's bar -- against the
    fp32 reference, or (truth64 given: products deep enough that the fp32 reference itself sits at the bar) arbitrated by
    float64 as tests/tolerance.close_arbitrated does: inside the bar around the true value, or no further from it than 1.5x
    the reference's own fp32 arithmetic."""
    got, want = got.detach().cpu().float(), want.detach().cpu().float()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    for name, fn in (("NaN", torch.isnan), ("+inf", torch.isposinf), ("-inf", torch.isneginf)):
        g, w = fn(got), fn(want)
        if not torch.equal(g, w):
            bad = (g != w).nonzero()
            first = tuple(int(v) for v in bad[0])
            raise AssertionError(f"{what}: {name} pattern differs at {bad.size(0)} of {g.numel()} elements "
                                 f"({int(w.sum())} expected); first at {first}: got {got[first].item()}, want {want[first].item()}")
    fin = torch.isfinite(want)
    if truth64 is not None:
        truth64 = truth64.detach().cpu().double()
        fin = fin & torch.isfinite(truth64)
    if int(fin.sum()) == 0:
        return

    def err(v, ref):
        d = (v[fin].double() - ref[fin].double()).abs()
        if norm:
            return float(d.max()) / max(1.0, float(ref[fin].double().abs().max()))
        return float((d / (1.0 + ref[fin].double().abs())).max())

    if truth64 is None:
        e, bar = err(got, want), TOL
    else:
        e, bar = err(got, truth64), max(TOL, 1.5 * err(want, truth64))
    assert e <= bar, f"{what}: finite elements off by {e:.3e} (bar {bar:.3e}, norm={norm})"


def plant_rows(x, g):
    """Special values at fixed places of a [n, k] operand (n >= 450): single elements, whole rows, both signs."""
    n, k = x.shape
    x[5, 3] = INF
    x[17, k - 1] = -INF
    x[33, 0] = NAN
    x[64, 7] = INF
    x[64, 9] = -INF                       # both infinities in one row: NaN wherever both weights are nonzero
    x[100, 5] = BIG
    x[101, 6] = -BIG
    x[130, 11] = HUGE
    x[131, 12] = -HUGE
    x[200] = torch.randn(k, generator=g) * TINY
    x[300] = torch.randn(k, generator=g) * SUB
    x[n - 1, 1] = INF                     # the ragged last tile
    return x


def plant_weight(w):
    """w [k, f]: weights that fit 8 mantissa bits (zero mid / lo pieces) and an exact zero, in the rows the planted inputs hit."""
    w[3, :8] = 0.5
    w[3, 8] = -0.25
    w[3, 9] = 0.0
    w[0, :4] = 1.0
    w[7, 2] = 0.0
    w[9, 2] = 0.0
    w[5] = w[5].clamp(-0.2, 0.2)          # BIG / HUGE times these stay finite
    w[6] = w[6].clamp(-0.2, 0.2)
    w[11] = w[11].clamp(-0.2, 0.2)
    w[12] = w[12].clamp(-0.2, 0.2)
    return w


TALL = [
    # rows, segment widths, f_out, transposed W, bias, output splits
    (4099, (128,), 64, True, False, None),              # C3a's input gradient shape
    (70001, (64,), 128, False, True, (64, 64)),         # C3a forward
    (2000, (64,), 192, False, True, (64, 64, 64)),      # C5a forward
    (2000, (64, 64, 64), 64, True, False, None),        # C5a's input gradient
    (451, (32,), 32, False, True, None),                # the smallest split shape
    (513, (256,), 64, False, False, None),              # eight k-blocks
]


@pytest.mark.parametrize("n,widths,f_out,transposed,with_bias,splits", TALL)
@pytest.mark.parametrize("where", ["x", "w", "both"])
def test_tall_product_default_form_carries_non_finite_values_as_torch_matmul(n, widths, f_out, transposed, with_bias, splits, where):
    from pytorch_geometric_signed_directed_amd.dense import set_tall_f32_exact, tall_product
    k = sum(widths)
    g = torch.Generator().manual_seed(n + k + f_out)
    x = torch.randn(n, k, generator=g)
    w = plant_weight(torch.randn(k, f_out, generator=g) / k ** 0.5)
    if where in ("x", "both"):
        x = plant_rows(x, g)
    if where in ("w", "both"):
        w[1, 5] = INF                       # a whole output column non-finite
        w[2, 6] = NAN
        w[4, f_out - 1] = -INF
    bias = torch.randn(f_out, generator=g) if with_bias else None
    with single_thread():
        want = x @ w
        if bias is not None:
            want = want + bias
    segs, at = [], 0
    for wd in widths:
        segs.append(x[:, at:at + wd].contiguous().to(dev()))
        at += wd
    wdev = (w.t().contiguous() if transposed else w).to(dev())
    prev = set_tall_f32_exact(False)
    try:
        out = tall_product(segs, wdev, transposed, None if bias is None else bias.to(dev()), splits=splits)
    finally:
        set_tall_f32_exact(prev)
    got = torch.cat([o.cpu() for o in out], dim=1) if splits is not None else out.cpu()
    same_special_values_and_close(got, want, f"tall product, non-finite in {where}")


def test_tall_product_finite_tiles_do_not_take_the_exact_branch():
    """The guard must not move finite results: the split form's output on finite inputs is bit-identical whether or not OTHER
    tiles of the same launch hold non-finite values."""
    from pytorch_geometric_signed_directed_amd.dense import set_tall_f32_exact, tall_product
    g = torch.Generator().manual_seed(11)
    n, k, f = 4096, 128, 64
    x = torch.randn(n, k, generator=g)
    w = (torch.randn(k, f, generator=g) / k ** 0.5).to(dev())
    prev = set_tall_f32_exact(False)
    try:
        clean = tall_product([x.to(dev())], w).cpu()
        x2 = x.clone()
        x2[16:32, 3] = INF                  # exactly one 16-row tile
        dirty = tall_product([x2.to(dev())], w).cpu()
    finally:
        set_tall_f32_exact(prev)
    keep = torch.ones(n, dtype=torch.bool)
    keep[16:32] = False
    assert torch.equal(clean[keep], dirty[keep])
    assert torch.isfinite(clean).all() and not torch.isfinite(dirty[16:32]).all()


GRAM = [
    # rows, X widths, G widths (>= 10 accumulator blocks: the 32x32 split form is the default there)
    (40003, (128,), (64, 64, 64)),
    (30000, (64,), (64, 64, 64)),
    (999, (64, 64), (192,)),
]


@pytest.mark.parametrize("n,xw,gw", GRAM)
@pytest.mark.parametrize("force32", [False, True])
def test_tall_gram_default_form_carries_non_finite_values_as_torch_matmul(n, xw, gw, force32, monkeypatch):
    from pytorch_geometric_signed_directed_amd.dense import set_tall_f32_exact, tall_gram
    if force32:
        monkeypatch.setenv("PYGSD_GRAM_32X32", "1")          # every 32-column shape through the split form
    g = torch.Generator().manual_seed(n + sum(xw) + sum(gw))
    x = plant_rows(torch.randn(n, sum(xw), generator=g), g)
    gm = torch.randn(n, sum(gw), generator=g)
    gm[5, 2] = 0.0                       # inf x 0 = NaN in the reference too
    gm[5, 3] = 0.5                       # fits 8 mantissa bits
    gm[17] = (gm[17] * 4).round() / 4
    gm[40, 7] = -INF
    gm[41, 8] = NAN
    gm[100] = gm[100].clamp(-0.2, 0.2)
    gm[101] = gm[101].clamp(-0.2, 0.2)
    gm[130] = gm[130].clamp(-0.2, 0.2)
    gm[131] = gm[131].clamp(-0.2, 0.2)
    with single_thread():
        want = x.t() @ gm
    xs, at = [], 0
    for wd in xw:
        xs.append(x[:, at:at + wd].contiguous().to(dev()))
        at += wd
    gs, at = [], 0
    for wd in gw:
        gs.append(gm[:, at:at + wd].contiguous().to(dev()))
        at += wd
    prev = set_tall_f32_exact(False)
    try:
        got = tall_gram(xs, gs)
    finally:
        set_tall_f32_exact(prev)
    same_special_values_and_close(got, want, "tall gram", norm=True)


DENSE = [(64, 64, 2, 1000), (64, 64, 2, 37), (128, 128, 3, 1030), (64, 128, 2, 600), (128, 64, 1, 451)]


@pytest.mark.parametrize("f_in,f_out,k1,n", DENSE)
@pytest.mark.parametrize("where", ["terms", "weight", "gradient"])
def test_magnetic_dense_default_form_carries_non_finite_values_as_the_reference_sequence(f_in, f_out, k1, n, where):
    """The dense stage of MagNetConv / MSConv and its gradients, the reference's way in fp32 on the CPU (matmul chains, then -, +,
    += bias: nn/directed/MagNetConv.py:189-247; gradients = what autograd derives from that sequence)."""
    from pytorch_geometric_signed_directed_amd.dense import dense_bwd_raw, dense_fwd_raw, set_dense_f32_exact
    g = torch.Generator().manual_seed(1000 * f_in + 10 * f_out + k1 + n)
    a = [torch.randn(n, f_in, generator=g) for _ in range(k1)]
    b = [torch.randn(n, f_in, generator=g) for _ in range(k1)]
    w = torch.randn(k1, f_in, f_out, generator=g) * 0.3
    for k in range(k1):
        w[k] = plant_weight(w[k])
    bias = torch.randn(f_out, generator=g)
    gr, gi = torch.randn(n, f_out, generator=g), torch.randn(n, f_out, generator=g)
    if where == "terms":
        a[0][5, 3] = INF
        b[k1 - 1][6, 3] = -INF
        a[k1 - 1][7, 0] = NAN
        if n > 20:
            a[0][17, 7] = INF
            b[0][17, 7] = INF             # A - B = NaN, A + B = inf at the same place
            a[0][18, 9] = INF
            b[0][18, 9] = -INF
            b[0][20, 5] = BIG
            a[k1 - 1][21, 6] = -HUGE
        a[0][n - 1, 1] = -INF
        gr[5, 2] = 0.0
        gi[5, 2] = 0.0                    # P = M = 0 against the inf: NaN in dW there, as in the reference
    elif where == "weight":
        w[0, 4, 7] = INF
        w[k1 - 1, 2, f_out - 1] = NAN
    else:
        gr[3, 1] = INF
        gi[4, 2] = -INF
        gr[9, 5] = NAN
        gr[n - 1, 0] = INF
        gi[n - 1, 0] = INF                # M = inf - inf = NaN, P = inf
    def reference(dtype):
        at = [t.detach().clone().to(dtype).requires_grad_() for t in a]
        bt = [t.detach().clone().to(dtype).requires_grad_() for t in b]
        wt, biast = w.detach().clone().to(dtype).requires_grad_(), bias.detach().clone().to(dtype).requires_grad_()
        with single_thread():
            rr = sum(at[k] @ wt[k] for k in range(k1))
            ii = sum(bt[k] @ wt[k] for k in range(k1))
            out_r, out_i = rr - ii + biast, rr + ii + biast
            torch.autograd.backward([out_r, out_i], [gr.to(dtype), gi.to(dtype)])
        return out_r, out_i, at, bt, wt, biast

    want_r, want_i, at, bt, wt, biast = reference(torch.float32)
    t_r, t_i, at64, bt64, wt64, biast64 = reference(torch.float64)
    d = dev()
    prev = set_dense_f32_exact(False)
    try:
        o_r, o_i = dense_fwd_raw([t.to(d) for t in a], [t.to(d) for t in b], w.to(d), bias.to(d))
        da, db, dw, dbias = dense_bwd_raw([t.to(d) for t in a], [t.to(d) for t in b], w.to(d), gr.to(d), gi.to(d))
    finally:
        set_dense_f32_exact(prev)
    same_special_values_and_close(o_r, want_r, f"dense out_real ({where})", truth64=t_r)
    same_special_values_and_close(o_i, want_i, f"dense out_imag ({where})", truth64=t_i)
    for k in range(k1):
        same_special_values_and_close(da[k], at[k].grad, f"dense dA_{k} ({where})", truth64=at64[k].grad)
        same_special_values_and_close(db[k], bt[k].grad, f"dense dB_{k} ({where})", truth64=bt64[k].grad)
    same_special_values_and_close(dw, wt.grad, f"dense dW ({where})", norm=True, truth64=wt64.grad)
    same_special_values_and_close(dbias, biast.grad, f"dense dbias ({where})", norm=True, truth64=biast64.grad)


def test_magnetic_dense_broadcast_gradient_row_with_non_finite_terms():
    """The gradient of a sum over the nodes arrives as ONE broadcast row (ldg = 0): the exact recomputation reads it the same way."""
    from pytorch_geometric_signed_directed_amd.dense import dense_bwd_raw, set_dense_f32_exact
    g = torch.Generator().manual_seed(3)
    n, f_in, f_out, k1 = 900, 64, 64, 2
    a = [torch.randn(n, f_in, generator=g) for _ in range(k1)]
    b = [torch.randn(n, f_in, generator=g) for _ in range(k1)]
    a[1][44, 3] = INF
    b[0][45, 60] = NAN
    w = torch.randn(k1, f_in, f_out, generator=g) * 0.3
    row_r, row_i = torch.randn(1, f_out, generator=g), torch.randn(1, f_out, generator=g)
    p32, m32 = (row_r + row_i).expand(n, f_out), (row_i - row_r).expand(n, f_out)
    with single_thread():
        dw32 = torch.stack([a[k].t() @ p32 + b[k].t() @ m32 for k in range(k1)])
    d = dev()
    prev = set_dense_f32_exact(False)
    try:
        _, _, dw, _ = dense_bwd_raw([t.to(d) for t in a], [t.to(d) for t in b], w.to(d), row_r.to(d).expand(n, f_out),
                                    row_i.to(d).expand(n, f_out))
    finally:
        set_dense_f32_exact(prev)
    same_special_values_and_close(dw, dw32, "dense dW, broadcast gradient row", norm=True)


@pytest.mark.parametrize("layer", ["magnet", "sgcn", "digcn"])
def test_layers_propagate_a_non_finite_feature_as_the_oracle(layer):
    """One +inf and one NaN input feature through a whole layer forward (operator product, then the dense stage) against the
    oracle's reference op sequence: the same rows / columns turn non-finite, the rest stays within the bar."""
    from oracle import ref_layers as R
    g = torch.Generator().manual_seed(17)
    n, e, f = 3000, 24000, 64
    ei = torch.randint(0, n, (2, e), generator=g)
    x = torch.randn(n, f, generator=g)
    x[7, 5] = INF
    x[11, 9] = NAN
    d = dev()
    if layer == "magnet":
        from pytorch_geometric_signed_directed_amd.nn import MagNetConv
        xi = torch.randn(n, f, generator=g)
        torch.manual_seed(0)
        conv = MagNetConv(f, f, 1, 0.25, False)
        op = R.magnet_operator(ei, None, n, 0.25, "sym", 2.0)
        want = R.magnet_conv(x, xi, op, conv.weight.detach(), conv.bias.detach())
        conv.to(d)
        got = conv(x.to(d), xi.to(d), ei.to(d))
        for o, wv, nm in zip(got, want, ("real", "imag")):
            same_special_values_and_close(o, wv, f"MagNetConv out_{nm}")
    elif layer == "sgcn":
        from pytorch_geometric_signed_directed_amd.nn.signed.SGCNConv import SGCNConv
        torch.manual_seed(0)
        conv = SGCNConv(f, f, first_aggr=True)
        pos, neg = ei[:, : e // 2], ei[:, e // 2:]
        want = R.sgcn_conv(x, pos, neg, (conv.lin_b.weight.detach(), conv.lin_b.bias.detach()),
                           (conv.lin_u.weight.detach(), conv.lin_u.bias.detach()), True, f)
        conv.to(d)
        got = conv(x.to(d), pos.to(d), neg.to(d))
        same_special_values_and_close(got, want, "SGCNConv")
    else:
        from pytorch_geometric_signed_directed_amd.nn.directed.DiGCNConv import DiGCNConv
        torch.manual_seed(0)
        conv = DiGCNConv(f, f)
        ew = torch.rand(e, generator=g)
        want = R.digcn_conv(x, ei, ew, conv.weight.detach(), conv.bias.detach())
        conv.to(d)
        got = conv(x.to(d), ei.to(d), ew.to(d))
        same_special_values_and_close(got, want, "DiGCNConv")


# ------------------------------------------------------------------ randomised planting
def _sprinkle(rng, t, count, big=False):
    """`count` special values at random places of a matrix: infinities of either sign, NaN, exact zeros, values that fit eight
    mantissa bits, tiny and subnormal ones; big: ONE magnitude next to the largest bf16 (3.3e38 below it, 3.4e38 above) as
    well.  One only, and its partners are kept below 0.2 by the caller: where several such terms meet in one sum, whether a
    PARTIAL sum passes FLT_MAX depends on the order of accumulation -- the reference's matmul does not define one (it
    computes rr and ii apart and overflows where rr - ii, accumulated together, does not)."""
    import numpy as np
    kinds = [INF, -INF, NAN, 0.0, 0.5, -0.25, TINY, SUB]
    n, k = t.shape
    for _ in range(count):
        t[int(rng.integers(0, n)), int(rng.integers(0, k))] = float(rng.choice(np.array(kinds)))
    if big:
        t[int(rng.integers(0, n)), int(rng.integers(0, k))] = float(rng.choice(np.array([BIG, -BIG, HUGE, -HUGE])))
    return t


def test_fuzz_non_finite_values_through_the_dense_products():
    """Random shapes (the kernels' own tile shapes and the generic fallbacks), random special values at random places of
    operands, weights and gradients, through tall_product (plain and transposed), tall_gram and the magnetic dense stage in
    their DEFAULT forms: NaN / +inf / -inf exactly where fp32 `torch.matmul` puts them on the CPU, the finite rest to the bar
    (float64 arbitrates).  PYGSD_FUZZ_ROUNDS / PYGSD_FUZZ_SEED as in tests/test_gpu_fuzz.py."""
    import os
    import numpy as np
    from pytorch_geometric_signed_directed_amd.dense import (dense_bwd_raw, dense_fwd_raw, dense_supported, set_dense_f32_exact,
                                                             set_tall_f32_exact, tall_gram, tall_product)
    rounds, seed0 = int(os.environ.get("PYGSD_FUZZ_ROUNDS", "6")), int(os.environ.get("PYGSD_FUZZ_SEED", "1000"))
    d = dev()
    widths = [4, 8, 16, 20, 32, 48, 64, 96, 128, 192, 256]
    for r in range(rounds):
        rng = np.random.default_rng(seed0 + 7 * r)
        normal = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))      # noqa: E731
        n = int(rng.integers(1, 6000)) if rng.random() < 0.8 else int(rng.integers(1, 40))
        # ---- tall product / gram
        segs = [int(rng.choice(widths[:9])) for _ in range(int(rng.integers(1, 4)))]
        k, f_out = sum(segs), int(rng.choice(widths))
        big = rng.random() < 0.5                                  # one near-FLT_MAX value in x; its partners stay below 0.2
        x, w = _sprinkle(rng, normal(n, k), int(rng.integers(0, 12)), big), normal(k, f_out) * 0.2
        up = normal(n, f_out)
        if big:
            w, up = w.clamp(-0.2, 0.2), up.clamp(-0.2, 0.2)
        w = _sprinkle(rng, w, int(rng.integers(0, 6)))
        up = _sprinkle(rng, up, int(rng.integers(0, 8)))
        bias = normal(f_out) if rng.random() < 0.5 else None
        tag = f"seed {seed0 + 7 * r} n={n} widths={segs}->{f_out}"
        with single_thread():
            want = x @ w if bias is None else x @ w + bias
            want_t, want_g = up @ w.t(), x.t() @ up
            t64 = x.double() @ w.double() if bias is None else x.double() @ w.double() + bias.double()
            t64_t, t64_g = up.double() @ w.double().t(), x.double().t() @ up.double()
        xs, at = [], 0
        for wd in segs:
            xs.append(x[:, at:at + wd].contiguous().to(d))
            at += wd
        prev = set_tall_f32_exact(False)
        try:
            got = tall_product(xs, w.to(d), False, None if bias is None else bias.to(d))
            got_t = tall_product([up.to(d)], w.to(d), True)
            got_g = tall_gram(xs, [up.to(d)])
        finally:
            set_tall_f32_exact(prev)
        same_special_values_and_close(got, want, tag + " product", truth64=t64)
        same_special_values_and_close(got_t, want_t, tag + " transposed product", truth64=t64_t)
        same_special_values_and_close(got_g, want_g, tag + " gram", norm=True, truth64=t64_g)
        # ---- magnetic dense stage
        f_in, f_o, k1 = int(rng.choice([64, 128])), int(rng.choice([64, 128])), int(rng.integers(1, 4))
        if not dense_supported(f_in, f_o, k1):
            continue
        a = [normal(n, f_in) for _ in range(k1)]
        b = [normal(n, f_in) for _ in range(k1)]
        big = int(rng.integers(0, 2 * k1)) if rng.random() < 0.5 else -1      # which of the 2 (K+1) term matrices gets the one
        for j, t in enumerate(a + b):
            _sprinkle(rng, t, int(rng.integers(0, 4)), j == big)
        wm, gr, gi = normal(k1, f_in, f_o) * 0.3, normal(n, f_o), normal(n, f_o)
        if big >= 0:
            wm, gr, gi = wm.clamp(-0.2, 0.2), gr.clamp(-0.1, 0.1), gi.clamp(-0.1, 0.1)
        for kk in range(k1):
            _sprinkle(rng, wm[kk], int(rng.integers(0, 3)))
        bm = normal(f_o)
        gr, gi = _sprinkle(rng, gr, int(rng.integers(0, 4))), _sprinkle(rng, gi, int(rng.integers(0, 4)))

        def reference(dtype):
            at_ = [t.detach().clone().to(dtype).requires_grad_() for t in a]
            bt_ = [t.detach().clone().to(dtype).requires_grad_() for t in b]
            wt_, bi_ = wm.detach().clone().to(dtype).requires_grad_(), bm.detach().clone().to(dtype).requires_grad_()
            with single_thread():
                rr = sum(at_[j] @ wt_[j] for j in range(k1))
                ii = sum(bt_[j] @ wt_[j] for j in range(k1))
                o_r, o_i = rr - ii + bi_, rr + ii + bi_
                torch.autograd.backward([o_r, o_i], [gr.to(dtype), gi.to(dtype)])
            return o_r, o_i, at_, bt_, wt_, bi_

        w32, w64 = reference(torch.float32), reference(torch.float64)
        prev = set_dense_f32_exact(False)
        try:
            o_r, o_i = dense_fwd_raw([t.to(d) for t in a], [t.to(d) for t in b], wm.to(d), bm.to(d))
            da, db, dw, dbias = dense_bwd_raw([t.to(d) for t in a], [t.to(d) for t in b], wm.to(d), gr.to(d), gi.to(d))
        finally:
            set_dense_f32_exact(prev)
        tag = f"seed {seed0 + 7 * r} n={n} dense {f_in}->{f_o} K+1={k1}"
        same_special_values_and_close(o_r, w32[0], tag + " out_real", truth64=w64[0])
        same_special_values_and_close(o_i, w32[1], tag + " out_imag", truth64=w64[1])
        for j in range(k1):
            same_special_values_and_close(da[j], w32[2][j].grad, tag + f" dA_{j}", truth64=w64[2][j].grad)
            same_special_values_and_close(db[j], w32[3][j].grad, tag + f" dB_{j}", truth64=w64[3][j].grad)
        same_special_values_and_close(dw, w32[4].grad, tag + " dW", norm=True, truth64=w64[4].grad)
        same_special_values_and_close(dbias, w32[5].grad, tag + " dbias", norm=True, truth64=w64[5].grad)
