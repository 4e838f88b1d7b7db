"""GPU: randomised differential checks of every layer on the hot path against the CPU oracle (oracle/ref_layers.py).

Every round draws a graph (size, density, self loops, duplicate entries, hub rows, isolated tails), feature widths that
are not the benchmark's (1 ... 150, odd ones included), layer options and upstream gradients from ONE seed, runs the
oracle's reference op sequence in fp32 and in float64 on the host and the HIP path on the GPU, and holds outputs and every
gradient to the bar of tests/tolerance.py with float64 as the arbiter (`close_arbitrated`): inside 1e-5 of the true value,
or -- where fp32 itself cannot be -- no worse than 3 x the reference sequence's own distance from it.

Why 3 and not the suite's 1.5, and why a rate: both fp32 results sit a RANDOM distance from float64 (hub rows of 10^4
entries summed in some order, hop chains, sums over all N rows), and each check takes the maximum over all elements of two
such draws.  Over tens of thousands of random checks the ratio of the two maxima passes 1.5 a few times per thousand and 3
a few times per ten thousand with nothing wrong (profiles/ holds the counts of a long run), so a check beyond 3 x is
logged as an outlier and a target fails on a GROSS one (50 x further: a defect) or on more than 1 outlier per 200 checks
(a systematic loss).  Both kinds were found this way: an own-feature block dropped for an edgeless operator (errors of
10^-1 ... 10^12), and a softmax backward whose row sums did not cancel as the reference's do (4 x the reference's error in
every round with 1 - 2 entries per row: 5 % of the checks).  A logged round is re-run by its seed:

        PYGSD_FUZZ_EXACT_SEED=<seed> python -m pytest tests/test_gpu_fuzz.py -k <target>

PYGSD_FUZZ_ROUNDS (default 6 per target, so the suite stays short) sets the length; profiles/ holds the record of a long
run.  A case whose float64 reference is itself non-finite (a degenerate normalisation) is skipped and counted -- the
non-finite contract of the products is tests/test_gpu_nonfinite.py's subject.
"""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import ref_layers as R
from tolerance import TOL, close_arbitrated, errors, single_thread

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")
ROUNDS = int(os.environ.get("PYGSD_FUZZ_ROUNDS", "6"))
SEED0 = int(os.environ.get("PYGSD_FUZZ_SEED", "1000"))
MAX_N = int(os.environ.get("PYGSD_FUZZ_MAX_NODES", "2500"))
MAX_E = int(os.environ.get("PYGSD_FUZZ_MAX_EDGES", "60000"))
SKIPPED = {"non_finite_reference": 0}
STATS = {}


# ------------------------------------------------------------------ drawing cases
def width(rng, top=150):
    """Feature widths: the kernels' vector widths and their neighbours as often as anything else."""
    pool = [1, 2, 3, 4, 5, 7, 8, 12, 15, 16, 17, 20, 24, 31, 32, 33, 48, 63, 64, 65, 72, 96, 100, 128, 129, 150]
    pool = [w for w in pool if w <= top]
    return int(rng.choice(pool)) if rng.random() < 0.7 else int(rng.integers(1, top + 1))


def draw_graph(rng, n_lo=1):
    """[2, E] int64 of a random digraph on n nodes: plain / one hub row and one hub column / duplicated entries /
    an isolated tail / no edges at all."""
    n = int(rng.integers(n_lo, MAX_N + 1)) if rng.random() < 0.8 else int(rng.integers(n_lo, 40))
    if rng.random() < 0.04:
        n = int(rng.integers(n_lo, n_lo + 3))             # one to three nodes: where "contiguous" stops implying anything
    density = float(rng.choice([0.0, 0.5, 2.0, 8.0, 30.0]))
    e = int(min(n * density, MAX_E))
    if density and rng.random() < 0.2:
        e = int(rng.integers(1, 8))
    src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    style = int(rng.integers(0, 5))
    if e and style == 1:                                  # a row and a column far longer than a wavefront's 64 slots
        k = int(rng.integers(1, e + 1))
        dst[:k] = rng.integers(0, n)
        src[e - min(k, e // 2):] = rng.integers(0, n)
    elif e > 1 and style == 2:                            # duplicate entries (the reference sums them)
        h = e // 2
        src[h:2 * h], dst[h:2 * h] = src[:h], dst[:h]
    elif e and style == 3 and n > 2:                      # nodes nothing touches
        hi = int(rng.integers(1, n))
        src, dst = src % hi, dst % hi
    elif e and style == 4:                                # self loops among the entries
        k = int(rng.integers(1, e + 1))
        dst[:k] = src[:k]
    return n, torch.from_numpy(np.stack([src, dst]).astype(np.int64))


def normal(rng, *shape):
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


def positive(rng, e):
    return torch.from_numpy((rng.random(e) + 0.25).astype(np.float32))


def finite(tensors):
    return all(bool(torch.isfinite(t).all()) for t in tensors.values())


def cast(t, dtype, device=None):
    if t is None:
        return None
    return t.to(dtype=dtype if t.is_floating_point() else t.dtype, device=device)


def rounds(target):
    """(seed, rng) of every round of one target; seeds differ across targets."""
    if os.environ.get("PYGSD_FUZZ_EXACT_SEED"):              # one logged round again
        seed = int(os.environ["PYGSD_FUZZ_EXACT_SEED"])
        yield seed, np.random.default_rng(seed)
        return
    base = SEED0 + 100003 * (sum(ord(c) for c in target) % 977)
    for r in range(ROUNDS):
        yield base + r, np.random.default_rng(base + r)


def run_rounds(target, one_round):
    """one_round(rng) -> (description, got, ref32, ref64, keys held to the max-norm bar).

    A check whose error exceeds max(1e-5, 3 x the reference sequence's) is an OUTLIER (logged with its seed).  The target fails
    on one GROSS outlier (beyond 50 x that bar: a defect, not rounding) or on more outliers than 1 in 200 checks (a
    systematic loss of accuracy shows up in every round of some kind of graph, a tail event of two fp32 roundings does not)."""
    outliers, gross = [], []
    st = STATS.setdefault(target, {"rounds": 0, "checks": 0, "above_1e-5": 0, "above_suite_bar_1.5x": 0, "outliers_3x": 0,
                                   "gross_outliers": 0, "worst_err_vs_float64": 0.0, "worst_ratio_to_reference_sequence": 0.0})
    for seed, rng in rounds(target):
        what, got, ref32, ref64, norm_keys = one_round(rng)
        if not finite(ref64):
            SKIPPED["non_finite_reference"] += 1
            continue
        st["rounds"] += 1
        for k in ref64:
            assert got.get(k) is not None, f"{target} seed={seed}: no {k} from the HIP path [{what}]"
            pick = 2 if k in norm_keys else 1
            mine, theirs = errors(got[k], ref64[k])[pick], errors(ref32[k], ref64[k])[pick]
            st["checks"] += 1
            st["above_1e-5"] += mine > TOL
            st["above_suite_bar_1.5x"] += mine > max(TOL, 1.5 * theirs)
            st["worst_err_vs_float64"] = max(st["worst_err_vs_float64"], mine)
            if mine > TOL:
                st["worst_ratio_to_reference_sequence"] = max(st["worst_ratio_to_reference_sequence"], mine / max(theirs, 1e-30))
            try:
                close_arbitrated(got[k], ref32[k], ref64[k], norm=k in norm_keys, what=f"{target} seed={seed} {k} [{what}]",
                                 slack=3.0, slack_ref=4.0)
            except AssertionError as err:                   # keep going: one run should list every bad seed
                line = f"seed {seed} {k} [{what}]: {err}{worst_element(got[k], ref32[k], ref64[k])}"
                outliers.append(line)
                if not mine <= 50.0 * max(TOL, 3.0 * theirs):
                    gross.append(line)
    st["outliers_3x"] += len(outliers)
    st["gross_outliers"] += len(gross)
    if outliers and os.environ.get("PYGSD_FUZZ_LOG"):
        with open(os.environ["PYGSD_FUZZ_LOG"], "a") as fh:
            fh.write("\n".join(outliers) + "\n")
    allowed = max(1, st["checks"] // 200)
    assert not gross, f"{len(gross)} gross mismatch(es) in {ROUNDS} rounds of {target}:\n" + "\n".join(gross[:20])
    assert len(outliers) <= allowed, (f"{len(outliers)} checks of {st['checks']} beyond 3 x the reference sequence's own error in "
                                      f"{ROUNDS} rounds of {target} (more than 1 in 200):\n" + "\n".join(outliers[:20]))


def worst_element(got, ref32, ref64):
    """Where the largest error sits, for the log: index, the three values there, the largest |value| of its row."""
    if got is None or got.numel() == 0:
        return ""
    g, a, b = got.detach().cpu().double(), ref32.detach().double(), ref64.detach()
    d = ((g - b).abs() / (1.0 + b.abs())).reshape(-1)
    i = int(torch.nan_to_num(d, nan=float("inf")).argmax())
    idx = tuple(int(v) for v in np.unravel_index(i, tuple(g.shape))) if g.dim() else ()
    row = b[idx[0]].abs().max().item() if g.dim() >= 2 else b.abs().max().item()
    return (f" | worst at {idx}: hip {g.reshape(-1)[i].item():.9g} fp32-ref {a.reshape(-1)[i].item():.9g} "
            f"float64 {b.reshape(-1)[i].item():.9g}, row max |.| {row:.4g}")


def grads(out, upstream, leaves):
    """{name: d(sum(out * upstream)) / d leaf}; a leaf the output does not depend on has a zero gradient."""
    out = out if isinstance(out, (tuple, list)) else (out,)
    upstream = upstream if isinstance(upstream, (tuple, list)) else (upstream,)
    total = sum((o * u.to(device=o.device, dtype=o.dtype)).sum() for o, u in zip(out, upstream))
    res = {f"out{i}": o.detach() for i, o in enumerate(out)}
    live = {k: v for k, v in leaves.items() if v is not None and v.requires_grad}
    if total.requires_grad and live:
        got = torch.autograd.grad(total, list(live.values()), allow_unused=True)
        for (k, leaf), g in zip(live.items(), got):
            res["d_" + k] = torch.zeros_like(leaf) if g is None else g
    return res


def leaf(t, dtype, device=None):
    return None if t is None else t.detach().to(dtype=dtype, device=device).clone().requires_grad_()


def leaf_dev(t, rng=None):
    """The product side's differentiable copy of an input.  Three times in ten a feature matrix arrives as the layers of a
    model hand it over: a COLUMN SLICE of a wider matrix (row stride > width, start not 16-byte aligned unless the left pad
    happens to be a multiple of 4) -- the gradient is then taken with respect to that view."""
    if t is None:
        return None
    t = t.detach().to(device=D, dtype=torch.float32)
    if t.dim() == 2 and rng is not None and rng.random() < 0.3:
        left, right = int(rng.integers(0, 6)), int(rng.integers(0, 6))
        wide = torch.cat([t.new_zeros(t.size(0), left), t, t.new_zeros(t.size(0), right)], 1).requires_grad_()
        return wide[:, left:left + t.size(1)]
    return t.clone().requires_grad_()


def three_ways(reference, product, tensors, params, upstream, rng=None):
    """reference(dtype, leaves) on the host in fp32 and float64, product(leaves) on the GPU; leaves = differentiable copies
    of `tensors` (inputs; with `rng`, sometimes as column slices: leaf_dev) and `params`."""
    res = []
    with single_thread():
        for dtype in (torch.float32, torch.float64):
            lv = {k: leaf(v, dtype) for k, v in {**tensors, **params}.items()}
            res.append(grads(reference(dtype, lv), [cast(u, dtype) for u in upstream], lv))
    lv = {k: leaf_dev(v, rng) for k, v in tensors.items()}
    lv.update({k: leaf(v, torch.float32, D) for k, v in params.items()})
    got = grads(product(lv), [u.to(D) for u in upstream], lv)
    return got, res[0], res[1]


# ------------------------------------------------------------------ structure work: bit-exact
def test_fuzz_structure_kernels_bit_exact():
    """COO -> CSR (stable by segment id: the reference's scatter order inside a row) and the key sort under every layer, on
    random sizes from empty to 3 M entries, skewed ids, rectangular shapes: row pointer, column ids and the permutation
    identical to a stable host sort."""
    from pytorch_geometric_signed_directed_amd.sparse import csr_from_coo
    from pytorch_geometric_signed_directed_amd.sparse_build import sort_keys
    for seed, rng in rounds("structure"):
        n_seg, n_src = int(10 ** rng.uniform(0, 5.5)), int(10 ** rng.uniform(0, 5.5))
        nnz = int(10 ** rng.uniform(0, 6.5)) if rng.random() < 0.9 else 0
        seg = rng.integers(0, n_seg, nnz)
        if nnz and rng.random() < 0.4:                              # skew: a few segments take most entries
            seg[:nnz // 2] = rng.integers(0, max(1, n_seg // 100), nnz // 2)
        src = rng.integers(0, n_src, nnz)
        seg_t, src_t = torch.from_numpy(seg.astype(np.int64)), torch.from_numpy(src.astype(np.int64))
        csr = csr_from_coo(seg_t.to(D), src_t.to(D), n_seg, n_src)
        order = torch.sort(seg_t, stable=True).indices
        want_ptr = torch.zeros(n_seg + 1, dtype=torch.long)
        want_ptr[1:] = torch.bincount(seg_t, minlength=n_seg).cumsum(0)
        tag = f"seed {seed} segments={n_seg} sources={n_src} nnz={nnz}"
        assert torch.equal(csr.rowptr.cpu().long(), want_ptr), tag
        assert torch.equal(csr.perm.cpu().long(), order), tag
        assert torch.equal(csr.col.cpu().long(), src_t[order]), tag
        bits = int(rng.integers(1, 41))
        m = int(10 ** rng.uniform(0, 6)) if rng.random() < 0.9 else 0
        keys = torch.from_numpy(rng.integers(0, 2 ** min(bits, 12), m).astype(np.int64))     # many duplicates: stability shows
        if bits > 12:
            keys = keys * (2 ** (bits - 12)) + torch.from_numpy(rng.integers(0, 2, m).astype(np.int64))
        skeys, perm = sort_keys(keys.to(D), bits)
        want = torch.sort(keys, stable=True)
        assert torch.equal(skeys.cpu(), want.values) and torch.equal(perm.cpu().long(), want.indices), f"seed {seed} sort m={m} bits={bits}"


# ------------------------------------------------------------------ the sparse products themselves
def test_fuzz_spmm_forward_backward():
    from pytorch_geometric_signed_directed_amd.sparse import Pattern, spmm

    def one(rng):
        n, ei = draw_graph(rng)
        n_in, n_out = n, n
        if rng.random() < 0.5 and ei.size(1):               # rectangular: ids stay inside both ranges
            n_in, n_out = int(ei[0].max()) + 1 + int(rng.integers(0, 5)), int(ei[1].max()) + 1 + int(rng.integers(0, 5))
        f, e = width(rng, 300 if rng.random() < 0.1 else 150), ei.size(1)
        weighted, reduce = rng.random() < 0.6, ("mean" if rng.random() < 0.3 else "add")
        flow = "target_to_source" if rng.random() < 0.3 else "source_to_target"
        if flow == "target_to_source":
            n_in, n_out = n_out, n_in
        with_z = reduce == "add" and rng.random() < 0.3
        alpha, beta = (2.0, -1.0) if with_z else (1.0, 0.0)
        x, w = normal(rng, n_in, f), (normal(rng, e) if weighted else None)
        z = normal(rng, n_out, f) if with_z else None
        up = [normal(rng, n_out, f)]

        def ref(dtype, lv):
            y = R.propagate(lv["x"], ei, lv["w"], n_out, flow=flow, reduce=reduce)
            return alpha * y + beta * lv["z"] if with_z else y

        def prod(lv):
            return spmm(Pattern(ei.to(D), n_in, n_out, flow), lv["x"], lv["w"], z=None if z is None else lv["z"].detach(),
                        alpha=alpha, beta=beta, reduce=reduce)

        got, r32, r64 = three_ways(ref, prod, {"x": x, "w": w, "z": z}, {}, up, rng)
        for d in (got, r32, r64):
            d.pop("d_z", None)                               # Z is an epilogue operand of the kernel, not differentiated
        return f"n={n_in}x{n_out} e={e} f={f} w={weighted} {reduce} {flow} z={with_z}", got, r32, r64, ()

    run_rounds("spmm", one)


def test_fuzz_spmm2_forward_backward():
    from pytorch_geometric_signed_directed_amd.sparse import Pattern, spmm2

    def one(rng):
        n, ei = draw_graph(rng)
        f, e = width(rng, 260 if rng.random() < 0.15 else 150), ei.size(1)
        t = {"xa": normal(rng, n, f), "xb": normal(rng, n, f), "wa": normal(rng, e), "wb": normal(rng, e)}
        up = [normal(rng, n, f), normal(rng, n, f)]

        def ref(dtype, lv):
            return R.propagate(lv["xa"], ei, lv["wa"], n), R.propagate(lv["xb"], ei, lv["wb"], n)

        def prod(lv):
            return spmm2(Pattern(ei.to(D), n, n), lv["xa"], lv["xb"], lv["wa"], lv["wb"])

        got, r32, r64 = three_ways(ref, prod, t, {}, up, rng)
        return f"n={n} e={e} f={f}", got, r32, r64, ()

    run_rounds("spmm2", one)


def test_fuzz_spmm_bf16_storage():
    """bf16 features (BASELINE config C5's storage): against the fp32 oracle on the bf16-ROUNDED inputs; on top of the
    accumulation bar only the final rounding of the result to bf16 (relative 2^-8) -- the bound of
    test_spmm_bf16_vs_oracle_on_rounded_inputs."""
    from pytorch_geometric_signed_directed_amd.sparse import Pattern, spmm
    bad = []
    for seed, rng in rounds("spmm_bf16"):
        n, ei = draw_graph(rng)
        f, e = width(rng, 260 if rng.random() < 0.15 else 150), ei.size(1)
        weighted, mean, with_z = rng.random() < 0.6, rng.random() < 0.3, rng.random() < 0.3
        x = normal(rng, n, f).to(torch.bfloat16)
        w = torch.from_numpy(rng.random(e).astype(np.float32)) if weighted else None
        z = normal(rng, n, f).to(torch.bfloat16) if with_z and not mean else None
        alpha, beta = (2.0, -1.0) if z is not None else (1.0, 0.0)
        with single_thread():
            want = alpha * R.propagate(x.double(), ei, None if w is None else w.double(), n, reduce="mean" if mean else "add")
            if z is not None:
                want = want + beta * z.double()
        got = spmm(Pattern(ei.to(D), n, n), x.to(D), None if w is None else w.to(D), z=None if z is None else z.to(D),
                   alpha=alpha, beta=beta, reduce="mean" if mean else "add")
        if got.dtype != torch.bfloat16:
            bad.append(f"seed {seed}: result dtype {got.dtype}")
            continue
        err = (got.double().cpu() - want).abs()
        bound = want.abs() * 2.0 ** -8 + 1e-5 * max(1.0, float(want.abs().max()) if want.numel() else 1.0)
        if not bool((err <= bound).all()):
            bad.append(f"seed {seed} n={n} e={e} f={f} w={weighted} mean={mean} z={z is not None}: worst excess {float((err - bound).max()):.3e}")
    assert not bad, "\n".join(bad[:20])


def test_fuzz_digcn_conv_bf16_storage():
    """DiGCNConv with bf16 parameters and features (BASELINE config C5's storage) at random widths -- the tiled bf16 shapes
    and everything the generic bf16 GEMM catches -- forward, input gradient and parameter gradients against the fp32 oracle
    on the bf16-rounded operands.  Every stored value (projection, product, gradient) is rounded to 8 bits of mantissa:
    relative 2^-8 of the quantity's scale per rounding, up to three roundings on the way to a gradient -- bound 3 * 2^-8 of
    the max norm (the bound of the sharded bf16 tests)."""
    from pytorch_geometric_signed_directed_amd.nn import DiGCNConv
    bad = []
    for seed, rng in rounds("digcn_bf16"):
        n, ei = draw_graph(rng)
        f_in, f_out, bias = width(rng), width(rng), rng.random() < 0.7
        w = (normal(rng, ei.size(1)) * 0.3)
        x, up = normal(rng, n, f_in), normal(rng, n, f_out)
        weight, b = normal(rng, f_in, f_out) * 0.3, (normal(rng, f_out) if bias else None)
        rnd = lambda t: None if t is None else t.to(torch.bfloat16).float()      # noqa: E731
        xo, wo, bo = rnd(x).requires_grad_(), rnd(weight).requires_grad_(), (None if b is None else rnd(b).requires_grad_())
        with single_thread():
            want = R.digcn_conv(xo, ei, w, wo, bo)
            (want * rnd(up)).sum().backward()
        layer = DiGCNConv(f_in, f_out, bias=bias)
        layer.load_state_dict({k: v for k, v in (("weight", weight), ("bias", b)) if v is not None})
        layer.to(D).to(torch.bfloat16)
        xd = x.to(D).to(torch.bfloat16).requires_grad_()
        out = layer(xd, ei.to(D), w.to(D))
        (out.float() * rnd(up).to(D)).sum().backward()
        pairs = [("out", out, want.detach()), ("dx", xd.grad, xo.grad), ("dweight", layer.weight.grad, wo.grad)]
        if bias:
            pairs.append(("dbias", layer.bias.grad, bo.grad))
        for name, got, ref in pairs:
            if got is None or got.dtype != torch.bfloat16:
                bad.append(f"seed {seed} {name}: {None if got is None else got.dtype}")
                continue
            err = float((got.float().cpu() - ref).abs().max()) if ref.numel() else 0.0
            scale = max(1.0, float(ref.abs().max())) if ref.numel() else 1.0
            if not err <= 3 * 2.0 ** -8 * scale:
                bad.append(f"seed {seed} n={n} e={ei.size(1)} {f_in}->{f_out} bias={bias} {name}: {err:.3e} > 3 * 2^-8 * {scale:.3g}")
    assert not bad, "\n".join(bad[:20])


# ------------------------------------------------------------------ a1 / a2: MagNetConv, MSConv
@pytest.mark.parametrize("signed", [False, True])
def test_fuzz_magnetic_layers(signed):
    from pytorch_geometric_signed_directed_amd.nn import MagNetConv, MSConv

    def one(rng):
        n, ei = draw_graph(rng, n_lo=2)
        e = ei.size(1)
        f_in, f_out, k = width(rng, 100), width(rng, 100), int(rng.choice([1, 1, 2, 3]))
        q = float(rng.choice([0.0, 0.25, 0.1, 0.5])) if rng.random() < 0.7 else float(rng.random() * 0.5)
        norm = "sym" if rng.random() < 0.7 else None
        bias, absdeg = rng.random() < 0.7, True
        if rng.random() < 0.25:
            w = None
        elif signed:
            w = positive(rng, e) * torch.from_numpy(rng.choice([-1.0, 1.0], e).astype(np.float32))
            if rng.random() < 0.3:
                w = torch.sign(w)                              # the +-1 build
        else:
            w = positive(rng, e)
        if norm == "sym":
            lam = None if rng.random() < 0.7 else float(1.5 + 2.0 * rng.random())
        else:
            # the unnormalised Laplacian's spectrum grows with the degrees: a lambda_max the caller would actually pass (the
            # reference's default is the largest eigenvalue, get_magnetic_Laplacian.py:88-92) -- here a Gershgorin bound on
            # it, so that 2 L / lambda_max - I stays a contraction and T_k does not grow like (degree)^k
            _, re, im = R.magnetic_laplacian(ei, w, n, q, None, signed, absdeg)
            idx = R.magnetic_laplacian(ei, w, n, q, None, signed, absdeg)[0][0]
            radius = torch.zeros(n).index_add_(0, idx, (re * re + im * im).sqrt())
            lam = float(max(radius.max().item(), 1.0) * (1.0 + 0.5 * rng.random()))
        # a trainable q (through the phase and the SDDMM edge-value gradients; the reference refuses it without normalisation,
        # MagNetConv.py:58-59) and an edge_weight that requires grad (the differentiable Laplacian build)
        train_q = norm == "sym" and q > 0.0 and rng.random() < 0.25
        train_w = w is not None and e > 0 and rng.random() < 0.3
        if signed:
            layer = MSConv(f_in, f_out, k, q, train_q, normalization=norm, bias=bias, absolute_degree=absdeg)
        else:
            layer = MagNetConv(f_in, f_out, k, q, train_q, normalization=norm, bias=bias)
        params = {"weight": normal(rng, k + 1, f_in, f_out) * 0.3, "bias": normal(rng, f_out) if bias else None}
        state = {kk: v for kk, v in params.items() if v is not None}
        if train_q:
            # every forward replaces the parameter by Parameter(clamp(q, 0, 0.25)) (MagNetConv.py:141-142): the oracle gets the
            # clamped value, and the gradient is that of the parameter the layer holds AFTER the forward
            state["q"] = torch.tensor([q], dtype=torch.float32)
            params["q"] = state["q"].clamp(0, 0.25)
        if train_w:
            params["w_edge"] = w
        layer.load_state_dict(state)
        layer.to(D)
        t = {"xr": normal(rng, n, f_in), "xi": normal(rng, n, f_in)}
        up = [normal(rng, n, f_out), normal(rng, n, f_out)]

        def ref(dtype, lv):
            op = R.magnet_operator(ei, lv["w_edge"] if train_w else cast(w, dtype), n, lv["q"] if train_q else q, norm,
                                   2.0 if lam is None else lam, signed, absdeg, dtype)
            return R.magnet_conv(lv["xr"], lv["xi"], op, lv["weight"], lv["bias"], duplicate=False)

        # parameter gradients come from the module's own parameters on the product side
        res = []
        with single_thread():
            for dtype in (torch.float32, torch.float64):
                lv = {kk: leaf(v, dtype) for kk, v in {**t, **params}.items()}
                res.append(grads(ref(dtype, lv), [cast(u, dtype) for u in up], lv))
        lv = {kk: leaf_dev(v, rng) for kk, v in t.items()}
        lv.update(weight=layer.weight, bias=layer.bias if bias else None)
        if train_w:
            lv["w_edge"] = leaf(w, torch.float32, D)
        w_dev = lv["w_edge"] if train_w else (None if w is None else w.to(D))
        out = layer(lv["xr"], lv["xi"], ei.to(D), w_dev, lambda_max=lam)
        if train_q:
            lv["q"] = layer.q
        got = grads(out, [u.to(D) for u in up], lv)
        what = (f"n={n} e={e} {f_in}->{f_out} K={k} q={q:.3f} norm={norm} lam={lam} bias={bias} w={'none' if w is None else 'yes'} "
                f"train_q={train_q} train_w={train_w}")
        return what, got, res[0], res[1], ("d_weight", "d_bias", "d_q")

    run_rounds("msconv" if signed else "magnet", one)


# ------------------------------------------------------------------ a5 / a7 / a8
def test_fuzz_digcn_conv():
    from pytorch_geometric_signed_directed_amd.nn import DiGCNConv

    def one(rng):
        n, ei = draw_graph(rng)
        f_in, f_out, bias = width(rng), width(rng), rng.random() < 0.7
        w = normal(rng, ei.size(1)) * 0.5                       # DiGCN's operator comes precomputed: any real values
        layer = DiGCNConv(f_in, f_out, bias=bias)
        params = {"weight": normal(rng, f_in, f_out) * 0.3, "bias": normal(rng, f_out) if bias else None}
        layer.load_state_dict({k: v for k, v in params.items() if v is not None})
        layer.to(D)
        x, up = normal(rng, n, f_in), [normal(rng, n, f_out)]
        res = []
        with single_thread():
            for dtype in (torch.float32, torch.float64):
                lv = {k: leaf(v, dtype) for k, v in {"x": x, **params}.items()}
                res.append(grads(R.digcn_conv(lv["x"], ei, cast(w, dtype), lv["weight"], lv["bias"]), [cast(up[0], dtype)], lv))
        lv = {"x": leaf_dev(x, rng), "weight": layer.weight, "bias": layer.bias if bias else None}
        got = grads(layer(lv["x"], ei.to(D), w.to(D)), [up[0].to(D)], lv)
        return f"n={n} e={ei.size(1)} {f_in}->{f_out} bias={bias}", got, res[0], res[1], ("d_weight", "d_bias")

    run_rounds("digcn", one)


@pytest.mark.parametrize("which", ["dgcn", "conv_base"])
def test_fuzz_normalised_propagates(which):
    from pytorch_geometric_signed_directed_amd.nn import Conv_Base, DGCNConv

    def one(rng):
        n, ei = draw_graph(rng)
        f = width(rng)
        w = positive(rng, ei.size(1)) if rng.random() < 0.6 else None
        loops, normalize = rng.random() < 0.7, rng.random() < 0.85
        if not normalize and w is None:
            w = positive(rng, ei.size(1))
        x, up = normal(rng, n, f), [normal(rng, n, f)]
        if which == "dgcn":
            improved = rng.random() < 0.3
            layer = DGCNConv(improved=improved, add_self_loops=loops, normalize=normalize)
            ref = lambda dtype, lv: R.dgcn_conv(lv["x"], ei, cast(w, dtype), improved, loops, normalize)
            what = f"improved={improved}"
        else:
            fill = float(rng.choice([0.5, 0.0, 1.0, 0.3]))
            layer = Conv_Base(fill, add_self_loops=loops, normalize=normalize)
            ref = lambda dtype, lv: R.conv_base(lv["x"], ei, cast(w, dtype), fill, loops, normalize)
            what = f"fill={fill}"
        got, r32, r64 = three_ways(ref, lambda lv: layer(lv["x"], ei.to(D), None if w is None else w.to(D)), {"x": x}, {}, up, rng)
        return f"n={n} e={ei.size(1)} f={f} loops={loops} normalize={normalize} w={w is not None} {what}", got, r32, r64, ()

    run_rounds(which, one)


# ------------------------------------------------------------------ a9: SIMPA / DIMPA
@pytest.mark.parametrize("directed", [False, True])
def test_fuzz_simpa(directed):
    from pytorch_geometric_signed_directed_amd.nn import SIMPA

    def one(rng):
        n, ei_p = draw_graph(rng)
        ei_n = torch.from_numpy(rng.integers(0, n, (2, int(rng.integers(0, 4 * n + 1)))).astype(np.int64))
        f, hop, fill = width(rng, 100), int(rng.integers(1, 4)), float(rng.choice([0.5, 1.0, 0.2]))
        w_p = positive(rng, ei_p.size(1)) if rng.random() < 0.6 else None
        w_n = positive(rng, ei_n.size(1)) if rng.random() < 0.6 else None
        layer = SIMPA(hop, fill, directed)
        params = {k: (normal(rng, *v.shape) * 0.5 + 0.5) for k, v in layer.state_dict().items()}
        layer.load_state_dict(params)
        layer.to(D)
        t = {"x_p": normal(rng, n, f), "x_n": normal(rng, n, f)}
        if directed:
            t.update(x_pt=normal(rng, n, f), x_nt=normal(rng, n, f))
        up = [normal(rng, n, (4 if directed else 2) * f)]
        res = []
        with single_thread():
            for dtype in (torch.float32, torch.float64):
                lv = {k: leaf(v, dtype) for k, v in {**t, **params}.items()}
                out = R.simpa(ei_p, cast(w_p, dtype), ei_n, cast(w_n, dtype), lv["x_p"], lv["x_n"], lv, hop, fill, directed,
                              lv.get("x_pt"), lv.get("x_nt"))
                res.append(grads(out, [cast(up[0], dtype)], lv))
        lv = {k: leaf_dev(v, rng) for k, v in t.items()}
        lv.update(dict(layer.named_parameters()))
        dw = lambda v: None if v is None else v.to(D)
        out = layer(ei_p.to(D), dw(w_p), ei_n.to(D), dw(w_n), lv["x_p"], lv["x_n"], lv.get("x_pt"), lv.get("x_nt"))
        got = grads(out, [up[0].to(D)], lv)
        what = f"n={n} e+={ei_p.size(1)} e-={ei_n.size(1)} f={f} hop={hop} fill={fill} w=({w_p is not None},{w_n is not None})"
        return what, got, res[0], res[1], tuple("d_" + k for k in params)

    run_rounds("simpa_directed" if directed else "simpa", one)


def test_fuzz_dimpa():
    from pytorch_geometric_signed_directed_amd.nn import DIMPA

    def one(rng):
        n, ei = draw_graph(rng)
        f, hop, fill = width(rng, 100), int(rng.integers(1, 4)), float(rng.choice([0.5, 1.0, 0.2]))
        w = positive(rng, ei.size(1)) if rng.random() < 0.6 else None
        layer = DIMPA(hop, fill)
        params = {k: (normal(rng, *v.shape) * 0.5 + 0.5) for k, v in layer.state_dict().items()}
        layer.load_state_dict(params)
        layer.to(D)
        t, up = {"x_s": normal(rng, n, f), "x_t": normal(rng, n, f)}, [normal(rng, n, 2 * f)]
        res = []
        with single_thread():
            for dtype in (torch.float32, torch.float64):
                lv = {k: leaf(v, dtype) for k, v in {**t, **params}.items()}
                out = R.dimpa(lv["x_s"], lv["x_t"], ei, cast(w, dtype), lv["_w_s"], lv["_w_t"], hop, fill)
                res.append(grads(out, [cast(up[0], dtype)], lv))
        lv = {k: leaf_dev(v, rng) for k, v in t.items()}
        lv.update(dict(layer.named_parameters()))
        got = grads(layer(lv["x_s"], lv["x_t"], ei.to(D), None if w is None else w.to(D)), [up[0].to(D)], lv)
        return f"n={n} e={ei.size(1)} f={f} hop={hop} fill={fill} w={w is not None}", got, res[0], res[1], ("d__w_s", "d__w_t")

    run_rounds("dimpa", one)


# ------------------------------------------------------------------ a10: SGCNConv
def test_fuzz_sgcn_conv():
    from pytorch_geometric_signed_directed_amd.nn import SGCNConv

    def one(rng):
        n, pos = draw_graph(rng)
        neg = torch.from_numpy(rng.integers(0, n, (2, int(rng.integers(0, 6 * n + 1)))).astype(np.int64))
        in_dim, out_dim = width(rng, 100), width(rng, 100)
        first, bias, norm_emb = rng.random() < 0.5, rng.random() < 0.7, rng.random() < 0.3
        layer = SGCNConv(in_dim, out_dim, first, bias=bias, norm_emb=norm_emb)
        params = {k: normal(rng, *v.shape) * 0.3 for k, v in layer.state_dict().items()}
        layer.load_state_dict(params)
        layer.to(D)
        x, up = normal(rng, n, in_dim if first else 2 * in_dim), [normal(rng, n, 2 * out_dim)]
        res = []
        with single_thread():
            for dtype in (torch.float32, torch.float64):
                lv = {k: leaf(v, dtype) for k, v in {"x": x, **params}.items()}
                out = R.sgcn_conv(lv["x"], pos, neg, (lv["lin_b.weight"], lv.get("lin_b.bias")),
                                  (lv["lin_u.weight"], lv.get("lin_u.bias")), first, in_dim, norm_emb)
                res.append(grads(out, [cast(up[0], dtype)], lv))
        lv = {"x": leaf_dev(x, rng)}
        lv.update(dict(layer.named_parameters()))
        got = grads(layer(lv["x"], pos.to(D), neg.to(D)), [up[0].to(D)], lv)
        hub = max([int(torch.bincount(e[0], minlength=1).max()) for e in (pos, neg) if e.size(1)] + [0])
        what = (f"n={n} e+={pos.size(1)} e-={neg.size(1)} {in_dim}->{out_dim} first={first} bias={bias} norm_emb={norm_emb} "
                f"max-out-degree={hub}")
        return what, got, res[0], res[1], tuple("d_" + k for k in params)

    run_rounds("sgcn", one)


# ------------------------------------------------------------------ a13: attention aggregate
def test_fuzz_gat_conv():
    from pytorch_geometric_signed_directed_amd.nn import GATConv

    def one(rng):
        n, ei = draw_graph(rng)
        f_in, f_out = width(rng, 64), width(rng, 48)
        heads, concat = int(rng.choice([1, 1, 2, 3])), rng.random() < 0.6
        bias, loops = rng.random() < 0.7, rng.random() < 0.8
        conv = GATConv(f_in, f_out, heads=heads, concat=concat, add_self_loops=loops, bias=bias)
        params = {k: normal(rng, *v.shape) * 0.3 for k, v in conv.state_dict().items()}
        conv.load_state_dict(params)
        conv.to(D)
        x, up = normal(rng, n, f_in), [normal(rng, n, heads * f_out if concat else f_out)]
        res = []
        with single_thread():
            for dtype in (torch.float32, torch.float64):
                lv = {k: leaf(v, dtype) for k, v in {"x": x, **params}.items()}
                out = R.gat_conv(lv["x"], ei, lv["lin.weight"], lv["att_src"], lv["att_dst"], lv.get("bias"), heads, concat,
                                 add_self_loops=loops)
                res.append(grads(out, [cast(up[0], dtype)], lv))
        lv = {"x": leaf_dev(x, rng)}
        lv.update(dict(conv.named_parameters()))
        got = grads(conv(lv["x"], ei.to(D)), [up[0].to(D)], lv)
        what = f"n={n} e={ei.size(1)} {f_in}->{f_out} heads={heads} concat={concat} bias={bias} loops={loops}"
        return what, got, res[0], res[1], tuple("d_" + k for k in params)

    run_rounds("gat", one)


def test_fuzz_snea_conv_toy_sizes():
    """SNEAConv (tanh attention over positive and negative incoming edges, target-row messages, partial self loops) against
    the node-by-node float64 formula of oracle/small_f64_torch.py -- plain Python per node, so graphs of <= 60 nodes; no fp32
    reference sequence exists at this level, the bar is the literal one against float64."""
    from oracle import small_f64_torch as F64
    from pytorch_geometric_signed_directed_amd.nn import SNEAConv
    from tolerance import close
    for seed, rng in rounds("snea"):
        n = int(rng.integers(2, 60))
        edges = lambda m: torch.from_numpy(rng.integers(0, n, (2, int(rng.integers(0, m)))).astype(np.int64))  # noqa: E731
        pos, neg = edges(5 * n), edges(3 * n)
        if rng.random() < 0.3 and pos.size(1):
            pos[1, :max(1, pos.size(1) // 2)] = int(rng.integers(0, n))         # one long row
        in_dim, out_dim, first = int(rng.integers(1, 12)), int(rng.integers(1, 12)), bool(rng.random() < 0.5)
        layer = SNEAConv(in_dim, out_dim, first)
        params = {k: normal(rng, *v.shape) * 0.4 for k, v in layer.state_dict().items()}
        layer.load_state_dict(params)
        layer.to(D)
        x, up = normal(rng, n, in_dim if first else 2 * in_dim), normal(rng, n, 2 * out_dim)
        p64 = {k: v.double().requires_grad_() for k, v in params.items()}
        x64 = x.double().requires_grad_()
        o64 = F64.snea_conv(x64, pos.numpy(), neg.numpy(), (p64["lin_b.weight"], p64["lin_b.bias"]), (p64["lin_u.weight"], p64["lin_u.bias"]),
                            (p64["alpha_b.weight"], p64["alpha_b.bias"]), (p64["alpha_u.weight"], p64["alpha_u.bias"]), first, in_dim)
        xd = x.to(D).requires_grad_()
        out = layer(xd, pos.to(D), neg.to(D))
        tag = f"snea seed={seed} n={n} e+={pos.size(1)} e-={neg.size(1)} {in_dim}->{out_dim} first={first}"
        close(out, o64.detach(), what=tag + " out")
        if not o64.requires_grad:                   # no edge and no re-added loop anywhere: the output is the constant zero
            assert float(out.detach().abs().max()) == 0.0, tag
            continue
        o64.backward(up.double())
        out.backward(up.to(D))
        close(xd.grad, torch.zeros_like(x64) if x64.grad is None else x64.grad, what=tag + " dx")
        for k, p in layer.named_parameters():
            want = p64[k].grad if p64[k].grad is not None else torch.zeros_like(p64[k])      # (a branch without any edge)
            close(torch.zeros_like(p) if p.grad is None else p.grad, want, norm=True, what=f"{tag} d {k}")


def test_fuzz_cut_and_imbalance_objectives():
    """SSSNET's three cut objectives and DIGRAC's imbalance objective (the per-cluster sparse mat-vecs as HIP
    SpMMs) on random signed / directed weighted graphs of <= 300 nodes and 2 ... 8 clusters against the dense float64
    formulas of oracle/small_f64_torch.py: values and the gradient of the probabilities, max-norm bar (they are sums over
    all nodes), literal 1e-5 against float64."""
    import scipy.sparse as sp
    from oracle import small_f64_torch as F64
    from pytorch_geometric_signed_directed_amd.utils import (Prob_Balanced_Normalized_Loss, Prob_Balanced_Ratio_Loss, Prob_Imbalance_Loss,
                                                             Unhappy_Ratio)
    from tolerance import close
    for seed, rng in rounds("losses"):
        n, k = int(rng.integers(4, 300)), int(rng.integers(2, 9))
        e = int(n * float(rng.choice([0.5, 3.0, 10.0])))
        r, c = rng.integers(0, n, e), rng.integers(0, n, e)
        prob0 = torch.softmax(normal(rng, n, k) * float(rng.choice([0.3, 2.0])), dim=1)
        tag = f"losses seed={seed} n={n} e={e} K={k}"
        # signed weights, stored as SSSNET stores them: positive and negative part of one scipy matrix
        w = rng.standard_normal(e)
        a = sp.coo_matrix((w, (r, c)), shape=(n, n)).tocsr()
        a_p, a_n = a.maximum(0), (-a).maximum(0)
        a_p.eliminate_zeros()
        a_n.eliminate_zeros()
        if (a_p - a_n).nnz:
            for i, cls in enumerate((Prob_Balanced_Normalized_Loss, Prob_Balanced_Ratio_Loss, Unhappy_Ratio)):
                prob = prob0.to(D).requires_grad_()
                val = cls(a_p, a_n)(prob)
                val.sum().backward()
                p64 = prob0.double().requires_grad_()
                v64 = F64.cut_losses(a_p.toarray(), a_n.toarray(), p64)[i]
                v64.backward()
                close(val.reshape(-1), v64.detach().reshape(-1), norm=True, what=f"{tag} {cls.__name__}")
                close(prob.grad, p64.grad, norm=True, what=f"{tag} d prob {cls.__name__}")
        # DIGRAC: nonnegative directed weights, every normalisation; gradients exist on the 'sort' branch only
        wd = rng.random(e) + 0.1
        dense = np.zeros((n, n))
        np.add.at(dense, (r, c), wd)
        ei = torch.from_numpy(np.stack([r, c]).astype(np.int64))
        adj = torch.sparse_coo_tensor(ei.to(D), torch.from_numpy(wd.astype(np.float32)).to(D), (n, n)).coalesce()
        pairs = k * (k - 1) // 2
        sel = int(rng.integers(1, pairs + 1))
        for norm in ("vol_sum", "vol_min", "vol_max", "plain"):
            for thr in ("sort", "std", "naive"):
                prob = prob0.to(D).requires_grad_()
                val = Prob_Imbalance_Loss(sel)(prob, adj, k, norm, thr)
                p64 = prob0.double().requires_grad_()
                v64 = F64.imbalance_loss(p64, dense, k, sel, norm, thr)
                close(val.reshape(-1), v64.detach().reshape(-1), norm=True, what=f"{tag} imbalance {norm} {thr} sel={sel}")
                if thr == "sort" and v64.requires_grad:
                    val.sum().backward()
                    v64.sum().backward()
                    close(prob.grad, p64.grad, norm=True, what=f"{tag} d prob imbalance {norm} sel={sel}")


# ------------------------------------------------------------------ the dense kernels at arbitrary widths
def test_fuzz_tall_products():
    from pytorch_geometric_signed_directed_amd import dense

    def one(rng):
        n = int(rng.integers(1, 20000)) if rng.random() < 0.8 else int(rng.integers(1, 70))
        segs = [width(rng, 130) for _ in range(int(rng.integers(1, 4)))]
        f_out, bias = width(rng, 200), rng.random() < 0.6
        xs = [normal(rng, n, s) for s in segs]
        wt, b = normal(rng, sum(segs), f_out) * 0.2, (normal(rng, f_out) if bias else None)
        up = [normal(rng, n, f_out)]
        t = {f"x{i}": x for i, x in enumerate(xs)}

        def ref(dtype, lv):
            y = torch.cat([lv[f"x{i}"] for i in range(len(segs))], 1) @ lv["w"]
            return y if lv["b"] is None else y + lv["b"]

        def prod(lv):
            return dense.tall_linear(torch.cat([lv[f"x{i}"] for i in range(len(segs))], 1), lv["w"], lv["b"])

        got, r32, r64 = three_ways(ref, prod, t, {"w": wt, "b": b}, up, rng)
        # the segmented entry points themselves (what the layers call: no concatenation), forward only
        # the segments as the layers hand them over: contiguous matrices, or column slices of one wider matrix (row stride)
        sliced = rng.random() < 0.4
        if sliced:
            wide = torch.cat([normal(rng, n, 3)] + xs + [normal(rng, n, 5)], 1).to(D)
            edges_ = np.cumsum([3] + segs)
            xd = [wide[:, int(edges_[i]):int(edges_[i + 1])] for i in range(len(segs))]
        else:
            xd = [x.to(D) for x in xs]
        got["segmented"] = dense.tall_product(xd, wt.to(D), False, None if b is None else b.to(D))
        # the weight gradient with the upstream gradient in 1 ... 3 column segments (SGCNConv: [g | g_a])
        cuts = sorted(set(int(c) for c in rng.integers(1, max(f_out, 2), int(rng.integers(0, 3))) if c < f_out))
        gd = up[0].to(D)
        gs = [gd[:, a:b_].contiguous() for a, b_ in zip([0] + cuts, cuts + [f_out])]
        got["gram"] = dense.tall_gram(xd, gs)
        got["transposed"] = dense.tall_product(gs, wt.to(D), True)            # dx = [g_0 | g_1 | ...] W^T
        for r, dtype in ((r32, torch.float32), (r64, torch.float64)):
            r["segmented"] = r["out0"]
            with single_thread():
                r["gram"] = torch.cat(xs, 1).to(dtype).t() @ up[0].to(dtype)
                r["transposed"] = up[0].to(dtype) @ wt.to(dtype).t()
        what = f"n={n} widths={segs}->{f_out} bias={bias} sliced={sliced} g-cuts={cuts}"
        return what, got, r32, r64, ("d_w", "d_b", "gram")

    run_rounds("tall", one)


def test_fuzz_two_forwards_before_one_backward():
    """One layer instance applied to two different inputs before a single backward of the summed objective (a layer shared
    between branches of a model; gradient accumulation over micro-batches with one backward): whatever the first forward
    saved for its backward must survive the second forward -- no saved tensor may live in a buffer the next call reuses.
    Input gradients must equal, bit for bit, those of two separate forward / backward passes; parameter gradients (one
    accumulation order against another) to the max-norm bar."""
    from pytorch_geometric_signed_directed_amd.nn import (DGCNConv, DIMPA, Conv_Base, DiGCNConv, GATConv, MagNetConv, MSConv,
                                                          SGCNConv, SIMPA, SNEAConv)
    from tolerance import close
    for seed, rng in rounds("twice"):
        n, ei = draw_graph(rng, n_lo=2)
        e = ei.size(1)
        ei2 = torch.from_numpy(rng.integers(0, n, (2, int(rng.integers(0, 4 * n + 1)))).astype(np.int64)).to(D)
        eid, w = ei.to(D), positive(rng, e).to(D)
        w2 = positive(rng, ei2.size(1)).to(D)
        f = int(rng.choice([4, 8, 12, 16, 20, 64]))
        kind = str(rng.choice(["magnet_k1", "magnet_k2", "msconv", "digcn", "dgcn", "conv_base", "simpa", "dimpa", "sgcn", "snea", "gat"]))
        torch.manual_seed(seed)
        two = kind in ("magnet_k1", "magnet_k2", "msconv", "simpa", "dimpa")          # layers that take two feature matrices
        if kind.startswith("magnet"):
            layer = MagNetConv(f, f, 1 if kind == "magnet_k1" else 2, 0.25, False).to(D)
            call = lambda a, b: layer(a, b, eid, w)                                   # noqa: E731
        elif kind == "msconv":
            layer = MSConv(f, f, 2, 0.1, False).to(D)
            call = lambda a, b: layer(a, b, eid, w)                                   # noqa: E731
        elif kind == "digcn":
            layer = DiGCNConv(f, f).to(D)
            call = lambda a, b: layer(a, eid, w)                                      # noqa: E731
        elif kind == "dgcn":
            layer = DGCNConv()
            call = lambda a, b: layer(a, eid, w)                                      # noqa: E731
        elif kind == "conv_base":
            layer = Conv_Base(0.5)
            call = lambda a, b: layer(a, eid, w)                                      # noqa: E731
        elif kind == "simpa":
            layer = SIMPA(2, 0.5, False).to(D)
            call = lambda a, b: layer(eid, w, ei2, w2, a, b)                          # noqa: E731
        elif kind == "dimpa":
            layer = DIMPA(2, 0.5).to(D)
            call = lambda a, b: layer(a, b, eid, w)                                   # noqa: E731
        elif kind == "sgcn":
            layer = SGCNConv(f, f, True).to(D)
            call = lambda a, b: layer(a, eid, ei2)                                    # noqa: E731
        elif kind == "snea":
            layer = SNEAConv(f, 8, True).to(D)
            call = lambda a, b: layer(a, eid, ei2)                                    # noqa: E731
        else:
            layer = GATConv(f, 8).to(D)
            call = lambda a, b: layer(a, eid)                                         # noqa: E731
        xs = [normal(rng, n, f).to(D) for _ in range(4)]

        def objective(out, salt):
            out = out if isinstance(out, (tuple, list)) else (out,)
            return sum((o * torch.sin(torch.arange(o.numel(), device=D, dtype=torch.float32).view_as(o) * (0.37 + salt) + k)).sum()
                       for k, o in enumerate(out))

        def leaves():
            return [x.clone().requires_grad_() for x in xs]

        params = list(layer.parameters()) if isinstance(layer, torch.nn.Module) else []
        # together: two forwards, one backward
        a = leaves()
        for p_ in params:
            p_.grad = None
        (objective(call(a[0], a[1]), 0.0) + objective(call(a[2], a[3]), 0.5)).backward()
        g_joint = [t.grad for t in a]
        p_joint = [None if p_.grad is None else p_.grad.clone() for p_ in params]
        # apart: forward / backward, forward / backward (parameter gradients accumulate)
        b = leaves()
        for p_ in params:
            p_.grad = None
        objective(call(b[0], b[1]), 0.0).backward()
        objective(call(b[2], b[3]), 0.5).backward()
        tag = f"twice seed={seed} {kind} n={n} e={e} f={f}"
        for i, (gj, ga) in enumerate(zip(g_joint, [t.grad for t in b])):
            if not two and i in (1, 3):
                continue
            assert (gj is None) == (ga is None), tag
            if gj is not None:
                assert torch.equal(gj, ga), f"{tag}: input gradient {i} differs by {float((gj - ga).abs().max()):.3e}"
        for pj, p_ in zip(p_joint, params):
            assert (pj is None) == (p_.grad is None), tag
            if pj is not None:
                close(pj, p_.grad, norm=True, what=tag + " parameter gradient")


def test_fuzz_hipgraph_replay_equals_eager():
    """A forward + backward of every layer captured into a hipGraph (hipgraph.capture_step) and replayed on NEW input values
    copied into the static tensors: outputs and input gradients bit for bit those of the eager call on the same values.
    (What a captured region requires is the caller's to provide: unchanged graph tensors -- the operator memo then keeps
    every host read out of the region -- and fixed shapes.)"""
    from pytorch_geometric_signed_directed_amd.hipgraph import capture_step
    from pytorch_geometric_signed_directed_amd.nn import (DGCNConv, DIMPA, Conv_Base, DiGCNConv, GATConv, MagNetConv, MSConv,
                                                          SGCNConv, SIMPA, SNEAConv)
    for seed, rng in rounds("hipgraph"):
        n, ei = draw_graph(rng, n_lo=2)
        eid, w = ei.to(D), positive(rng, ei.size(1)).to(D)
        ei2 = torch.from_numpy(rng.integers(0, n, (2, int(rng.integers(0, 4 * n + 1)))).astype(np.int64)).to(D)
        w2 = positive(rng, ei2.size(1)).to(D)
        f = int(rng.choice([4, 8, 16, 20, 64]))
        kind = str(rng.choice(["magnet_k1", "magnet_k2", "msconv", "digcn", "dgcn", "conv_base", "simpa", "dimpa", "sgcn", "snea", "gat"]))
        torch.manual_seed(seed)
        if kind.startswith("magnet"):
            layer = MagNetConv(f, f, 1 if kind == "magnet_k1" else 2, 0.25, False).to(D)
            call = lambda a, b: layer(a, b, eid, w)                                   # noqa: E731
        elif kind == "msconv":
            layer = MSConv(f, f, 2, 0.1, False).to(D)
            call = lambda a, b: layer(a, b, eid, w)                                   # noqa: E731
        elif kind == "digcn":
            layer = DiGCNConv(f, f).to(D)
            call = lambda a, b: layer(a, eid, w)                                      # noqa: E731
        elif kind == "dgcn":
            layer = DGCNConv()
            call = lambda a, b: layer(a, eid, w)                                      # noqa: E731
        elif kind == "conv_base":
            layer = Conv_Base(0.5)
            call = lambda a, b: layer(a, eid, w)                                      # noqa: E731
        elif kind == "simpa":
            layer = SIMPA(2, 0.5, False).to(D).requires_grad_(False)                  # (trainable hop weights are read on the host)
            call = lambda a, b: layer(eid, w, ei2, w2, a, b)                          # noqa: E731
        elif kind == "dimpa":
            layer = DIMPA(2, 0.5).to(D)
            call = lambda a, b: layer(a, b, eid, w)                                   # noqa: E731
        elif kind == "sgcn":
            layer = SGCNConv(f, f, True).to(D)
            call = lambda a, b: layer(a, eid, ei2)                                    # noqa: E731
        elif kind == "snea":
            layer = SNEAConv(f, 8, True).to(D)
            call = lambda a, b: layer(a, eid, ei2)                                    # noqa: E731
        else:
            layer = GATConv(f, 8).to(D)
            call = lambda a, b: layer(a, eid)                                         # noqa: E731
        a, b = normal(rng, n, f).to(D).requires_grad_(), normal(rng, n, f).to(D).requires_grad_()

        def step():
            a.grad = b.grad = None
            out = call(a, b)
            out = out if isinstance(out, (tuple, list)) else (out,)
            sum((o * o).sum() for o in out).backward()
            return tuple(out) + (a.grad, b.grad)

        tag = f"hipgraph seed={seed} {kind} n={n} e={ei.size(1)} f={f}"
        replay = capture_step(step, warmup=2)
        for _ in range(2):                                   # new values into the static inputs, then replay vs eager
            va, vb = normal(rng, n, f).to(D), normal(rng, n, f).to(D)
            with torch.no_grad():
                a.copy_(va)
                b.copy_(vb)
            got = [None if t is None else t.detach().clone() for t in replay()]
            torch.cuda.synchronize()
            a2, b2 = va.clone().requires_grad_(), vb.clone().requires_grad_()
            out = call(a2, b2)
            out = out if isinstance(out, (tuple, list)) else (out,)
            sum((o * o).sum() for o in out).backward()
            want = list(out) + [a2.grad, b2.grad]
            for i, (g_, w_) in enumerate(zip(got, want)):
                assert (g_ is None) == (w_ is None), tag
                if g_ is not None:
                    assert torch.equal(g_, w_.detach()), f"{tag}: tensor {i} differs by {float((g_ - w_).abs().max()):.3e}"


def test_fuzz_threads_and_streams_share_layers():
    """Four host threads, each on its own HIP stream, call the SAME layer instances on the SAME graph tensors in random order
    (forward + backward), with the memos cleared now and then by one of them: every result bit for bit the one computed
    alone beforehand.  Exercises the memo locks, the ordering of hits across streams, the thread-local error state of the
    C-ABI and the allocator-owned workspaces."""
    import threading
    from pytorch_geometric_signed_directed_amd import memo
    from pytorch_geometric_signed_directed_amd.nn import DGCNConv, Conv_Base, GATConv, MagNetConv, SGCNConv, SIMPA
    rng = np.random.default_rng(SEED0 + 77)
    n, f = 3000, 16
    ei = torch.from_numpy(rng.integers(0, n, (2, 40000)).astype(np.int64)).to(D)
    ei2 = torch.from_numpy(rng.integers(0, n, (2, 15000)).astype(np.int64)).to(D)
    w, w2 = positive(rng, 40000).to(D), positive(rng, 15000).to(D)
    xs = [normal(rng, n, f).to(D) for _ in range(3)]
    torch.manual_seed(5)
    layers = {"magnet": MagNetConv(f, f, 2, 0.25, False).to(D), "dgcn": DGCNConv(), "conv_base": Conv_Base(0.5),
              "simpa": SIMPA(2, 0.5, False).to(D).requires_grad_(False), "sgcn": SGCNConv(f, f, True).to(D), "gat": GATConv(f, 8).to(D)}
    calls = {"magnet": lambda m, a, b: torch.cat(m(a, b, ei, w), 1), "dgcn": lambda m, a, b: m(a, ei, w),
             "conv_base": lambda m, a, b: m(a, ei, w), "simpa": lambda m, a, b: m(ei, w, ei2, w2, a, b),
             "sgcn": lambda m, a, b: m(a, ei, ei2), "gat": lambda m, a, b: m(a, ei)}

    def run(kind, i, j):
        a, b = xs[i].clone().requires_grad_(), xs[j].clone().requires_grad_()
        out = calls[kind](layers[kind], a, b)
        (out * out).sum().backward(inputs=[a])          # (parameter gradients are not accumulated: the threads share the modules)
        return out.detach(), a.grad

    want = {(k, i, j): run(k, i, j) for k in layers for i in range(3) for j in range(3)}
    torch.cuda.synchronize()
    memo.clear_all()
    errors = []

    def worker(tid):
        try:
            trng = np.random.default_rng(SEED0 + 1000 + tid)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for it in range(6 * max(ROUNDS, 4)):
                    kind = str(trng.choice(list(layers)))
                    i, j = int(trng.integers(0, 3)), int(trng.integers(0, 3))
                    if tid == 0 and it % 7 == 3:
                        memo.clear_all()
                    out, grad = run(kind, i, j)
                    stream.synchronize()
                    ro, rg = want[(kind, i, j)]
                    if not (torch.equal(out, ro) and torch.equal(grad, rg)):
                        errors.append(f"thread {tid} iteration {it} {kind}: out {float((out - ro).abs().max()):.3e} grad {float((grad - rg).abs().max()):.3e}")
                        return
        except Exception as exc:                         # noqa: BLE001 -- reported by the main thread
            errors.append(f"thread {tid}: {type(exc).__name__}: {exc}")

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, "\n".join(errors[:10])


def test_fuzz_memo_never_serves_a_stale_operator():
    """The operator / pattern memos (memo.py: keyed on tensor identity, in-place version and storage) under random histories:
    one long-lived instance of every uncached layer is called again and again while its graph tensors are, at random, left
    alone, edited in place through torch (version bump), replaced by an equal copy, replaced by different content, freed
    and re-allocated (the storage may come back at the same address), moved through `.clone()` of a view ...; after every
    step the output must equal, BIT FOR BIT, that of a never-used copy of the layer called on clones of the graph tensors.  Writes that bypass the
    version counter (`.data`, raw pointers) are the documented exception of the default mode; under PYGSD_MEMO_VERIFY=1
    (strict mode: content fingerprints) the histories include them."""
    from pytorch_geometric_signed_directed_amd import memo
    from pytorch_geometric_signed_directed_amd.nn import DGCNConv, DIMPA, Conv_Base, GATConv, MagNetConv, MSConv, SGCNConv, SIMPA, SNEAConv

    def same(a, b):
        a = a if isinstance(a, (tuple, list)) else (a,)
        b = b if isinstance(b, (tuple, list)) else (b,)
        return all(torch.equal(x, y) for x, y in zip(a, b))

    bad = []
    for seed, rng in rounds("memo"):
        n = int(rng.integers(8, 600))

        def graph(m=None):
            e = int(n * float(rng.choice([1.0, 4.0, 12.0]))) if m is None else m
            return (torch.from_numpy(rng.integers(0, n, (2, e)).astype(np.int64)).to(D),
                    torch.from_numpy((rng.random(e) + 0.25).astype(np.float32)).to(D))

        f = int(rng.choice([4, 8, 16, 20]))
        torch.manual_seed(seed)
        kind = str(rng.choice(["magnet", "msconv", "dgcn", "conv_base", "simpa", "dimpa", "sgcn", "snea", "gat"]))
        state = {"ei": None, "w": None, "ei2": None, "w2": None, "x": normal(rng, n, f).to(D), "x2": normal(rng, n, f).to(D)}
        state["ei"], state["w"] = graph()
        state["ei2"], state["w2"] = graph()
        if kind == "magnet":
            layer = MagNetConv(f, f, 2, 0.25, False).to(D)
            call = lambda m, st: m(st["x"], st["x2"], st["ei"], st["w"])                                # noqa: E731
        elif kind == "msconv":
            layer = MSConv(f, f, 1, 0.1, False).to(D)
            call = lambda m, st: m(st["x"], st["x2"], st["ei"], st["w"])                                # noqa: E731
        elif kind == "dgcn":
            layer = DGCNConv()
            call = lambda m, st: m(st["x"], st["ei"], st["w"])                                    # noqa: E731
        elif kind == "conv_base":
            layer = Conv_Base(0.5)
            call = lambda m, st: m(st["x"], st["ei"], st["w"])                                    # noqa: E731
        elif kind == "simpa":
            layer = SIMPA(2, 0.5, False).to(D)
            call = lambda m, st: m(st["ei"], st["w"], st["ei2"], st["w2"], st["x"], st["x2"])           # noqa: E731
        elif kind == "dimpa":
            layer = DIMPA(2, 0.5).to(D)
            call = lambda m, st: m(st["x"], st["x2"], st["ei"], st["w"])                          # noqa: E731
        elif kind == "snea":
            layer = SNEAConv(f, 8, True).to(D)
            call = lambda m, st: m(st["x"], st["ei"], st["ei2"])                                  # noqa: E731
        elif kind == "sgcn":
            layer = SGCNConv(f, f, True).to(D)
            call = lambda m, st: m(st["x"], st["ei"], st["ei2"])                                  # noqa: E731
        else:
            layer = GATConv(f, 8).to(D)
            call = lambda m, st: m(st["x"], st["ei"])                                             # noqa: E731
        pristine = copy.deepcopy(layer)                    # never called: no history of its own (per-instance memos included)
        history = []
        for step in range(int(rng.integers(4, 10))):
            acts = ["same", "same", "edit_index", "edit_weight", "equal_copy", "new_graph", "realloc", "swap", "resize"]
            if memo.verify():                          # strict mode (PYGSD_MEMO_VERIFY=1): writes BEHIND the version counter too
                acts += ["data_edit_index", "data_edit_weight"]
            act = str(rng.choice(acts))
            which = "2" if (rng.random() < 0.3 and kind in ("simpa", "sgcn", "snea")) else ""
            ei, w = state["ei" + which], state["w" + which]
            if act == "edit_index" and ei.size(1):
                j = int(rng.integers(0, ei.size(1)))
                ei[int(rng.integers(0, 2)), j] = int(rng.integers(0, n))
            elif act == "edit_weight" and w.numel():
                w[int(rng.integers(0, w.numel()))] += 0.5
            elif act == "data_edit_index" and ei.size(1):
                ei.data[int(rng.integers(0, 2)), int(rng.integers(0, ei.size(1)))] = int(rng.integers(0, n))
            elif act == "data_edit_weight" and w.numel():
                w.data[int(rng.integers(0, w.numel()))] += 0.5
            elif act == "equal_copy":
                state["ei" + which], state["w" + which] = ei.clone(), w.clone()
            elif act == "new_graph":
                state["ei" + which], state["w" + which] = graph()
            elif act == "realloc":                         # free, then allocate the same shape again: often the same address
                shape = ei.size(1)
                state["ei" + which] = state["w" + which] = None
                del ei, w
                state["ei" + which], state["w" + which] = graph(shape)
            elif act == "swap" and kind in ("simpa", "sgcn", "snea"):
                state["ei"], state["ei2"] = state["ei2"], state["ei"]
                state["w"], state["w2"] = state["w2"], state["w"]
            elif act == "resize":                          # a slice of the same storage: same data pointer, fewer entries
                keep = max(1, state["ei" + which].size(1) // 2)
                state["ei" + which] = state["ei" + which][:, :keep]
                state["w" + which] = state["w" + which][:keep]
            history.append(act + which)
            # forward and backward (the transposed operator and its value arrays are memoised too): outputs and the input
            # gradient.  The reference: a never-used copy of the layer on CLONES of the graph tensors -- new identities, which
            # no memo (per-instance or module-level) can have seen; the process-wide switch is not touched (switching it off
            # clears every memo, which would also wipe what the long-lived layer is being tested for remembering)
            def run(m, st):
                st = dict(st, x=st["x"].detach().clone().requires_grad_())
                out = call(m, st)
                out = out if isinstance(out, (tuple, list)) else (out,)
                sum(o.sum() for o in out).backward()
                return tuple(o.detach() for o in out) + (st["x"].grad,)

            got = run(layer, state)
            want = run(copy.deepcopy(pristine), {k: v.clone() for k, v in state.items()})
            if not same(got, want):
                bad.append(f"seed {seed} {kind} n={n} f={f} after {history}")
                break
    assert not bad, "\n".join(bad[:20])


def test_fuzz_memo_check_would_see_a_memo_that_ignores_versions(monkeypatch):
    """The check above has teeth: with the in-place version taken out of the memo's key (what a memo keyed on identity and
    storage alone would be) the same histories must produce stale operators, and the check must say so."""
    from pytorch_geometric_signed_directed_amd import memo
    if memo.verify():
        pytest.skip("strict mode (PYGSD_MEMO_VERIFY=1) checks contents: it sees the edits whatever the key holds")
    real = memo._stamp
    monkeypatch.setattr(memo, "_stamp", lambda t: None if t is None else (0,) + tuple(real(t)[1:]))
    memo.clear_all()
    try:
        with pytest.raises(AssertionError, match="after"):
            test_fuzz_memo_never_serves_a_stale_operator()
    finally:
        memo.clear_all()


def test_fuzz_report():
    """Not a check: leaves the run's counts (per target: rounds, checks, how many sat above the literal 1e-5 bar, above the
    suite's 1.5 x bar, above this file's 3 x bar) in the output and, when PYGSD_FUZZ_LOG names a path, in <path>.json."""
    import json
    from pytorch_geometric_signed_directed_amd import _cabi
    routes = _cabi.library_routes()
    # every product of the rounds above was fp32 or bf16, at widths from 1 to 300: none may have reached hipBLASLt / rocBLAS
    assert not routes, f"dense products took a library route during the randomised rounds: {routes}"
    report = {"rounds_per_target": ROUNDS, "first_seed": SEED0, "max_nodes": MAX_N, "max_edges": MAX_E,
              "skipped_non_finite_reference": SKIPPED["non_finite_reference"], "targets": STATS}
    print(json.dumps(report, indent=1))
    if os.environ.get("PYGSD_FUZZ_LOG"):
        with open(os.environ["PYGSD_FUZZ_LOG"] + ".json", "w") as fh:
            json.dump(report, fh, indent=1)
