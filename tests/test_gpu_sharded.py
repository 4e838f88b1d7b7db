"""GPU: the node-sharded layers (HIP compute; row layout and p_r x p_c grid, pipelined in column phases and
return chunks) against the un-sharded oracle.  2 .. 8 ranks share the one GPU of the test box and exchange
through gloo (RCCL refuses two ranks on one device; the 8-GPU RCCL run is the driver's) -- plan, distributed
operator build, packing / slicing, partial products, the Chebyshev adjoint over exchanged blocks and the
parameter all-reduce are the production code.  BASELINE configs C4 (sharded signed MSConv, h = 128, K = 2) and
C5 (sharded DiGCN inception block, fp32 and bf16) are exercised here in their stated form at test size."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _errs(pairs):
    """(worst per-element |d| / (1 + |want|), worst max-norm-relative) over (got, want) pairs."""
    mixed = rel = 0.0
    for got, want in pairs:
        d = (got.double() - want.double()).abs()
        mixed = max(mixed, float((d / (1.0 + want.double().abs())).max()))
        rel = max(rel, float(d.max()) / max(1.0, float(want.abs().max())))
    return mixed, rel


def _magnetic_case(rank, world, cfg):
    from oracle import ref_layers as R
    from pytorch_geometric_signed_directed_amd.parallel import ShardedMagNetConv, all_gather_rows
    n, k, f, layout, phases, chunks, signed, absdeg, build = cfg
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    e = 15 * n
    ei = torch.randint(0, n, (2, e), generator=g)
    ei[:, :e // 3] = torch.randint(0, max(n // 8, 2), (2, e // 3), generator=g)      # skew: uneven balanced ranges
    w = torch.rand(e, generator=g) + 0.5
    if signed:
        w = w * (torch.randint(0, 2, (e,), generator=g) * 2 - 1)
    xr, xi = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
    gr, gi = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
    torch.manual_seed(11)
    layer = ShardedMagNetConv(f, f, k, 0.25, n, ei.to(dev), w.to(dev), device=dev, layout=layout, signed=signed,
                              absolute_degree=absdeg, phases=phases, return_chunks=chunks, build=build,
                              grid_cols=2 if (layout == "grid" and world == 2) else None)   # force the 1 x 2 grid
    assert layer.layout == layout and (layout == "rows" or layer.engine.p_c > 1)
    with torch.no_grad():
        layer.bias.uniform_(-0.5, 0.5)
        dist.broadcast(layer.bias.data, 0)
    a = layer.shard_rows(xr.to(dev)).requires_grad_()
    b = layer.shard_rows(xi.to(dev)).requires_grad_()
    o_r, o_i = layer(a, b)
    ((o_r * layer.shard_rows(gr.to(dev))).sum() + (o_i * layer.shard_rows(gi.to(dev))).sum()).backward()
    plan = layer.plan
    got = [plan.unshard_rows(all_gather_rows(t.detach().contiguous())).cpu() for t in (o_r, o_i, a.grad, b.grad)]
    # un-sharded oracle (reference op sequence, CPU)
    weight, bias = layer.weight.detach().cpu(), layer.bias.detach().cpu()
    c, d = xr.clone().requires_grad_(), xi.clone().requires_grad_()
    wt, bs = weight.clone().requires_grad_(), bias.clone().requires_grad_()
    op = R.magnet_operator(ei, w, n, 0.25, "sym", 2.0, signed=signed, absolute_degree=absdeg)
    w_r, w_i = R.magnet_conv(c, d, op, wt, bs, duplicate=False)
    ((w_r * gr).sum() + (w_i * gi).sum()).backward()
    rows = _errs(zip(got, [w_r.detach(), w_i.detach(), c.grad, d.grad]))[0]          # absolute bar
    prm = _errs([(layer.weight.grad.cpu(), wt.grad), (layer.bias.grad.cpu(), bs.grad)])[1]   # row reductions
    return rows, prm, layer.global_nnz, int(op[0].size(1)) - n


# n, K, f, layout, phases, return chunks, signed, absolute_degree, build -- grouped by world size: one process
# group per world runs all of its configurations (spawning ranks costs more than the layers do)
MAGNETIC = {
    2: [(1003, 2, 64, "rows", 2, 1, False, True, "distributed"),
        (300, 2, 6, "rows", 2, 1, False, True, "global"),               # width the vector kernel must pad
        (1003, 1, 64, "grid", 2, 2, False, True, "distributed"),        # 1 x 2
        # BASELINE C4: MSGNN's signed magnetic Laplacian (general/MSConv.py:121-230), h = 128, K = 2, node-partitioned
        (900, 2, 128, "rows", 2, 1, True, True, "distributed")],
    3: [(500, 3, 16, "rows", 1, 1, False, True, "distributed")],
    4: [(901, 3, 32, "grid", 2, 2, False, True, "distributed"),         # 1 x 4 at 8-float slices, three orders
        (900, 2, 128, "rows", 2, 1, True, False, "distributed"),        # C4
        (900, 2, 128, "grid", 2, 2, True, True, "distributed")],        # C4
    8: [(1203, 2, 64, "grid", 2, 2, False, True, "distributed"),        # 2 x 4: the 8-GPU configuration
        (1203, 1, 64, "grid", 1, 1, False, True, "global"),             # 2 x 4 un-pipelined, global build
        (1100, 2, 128, "rows", 2, 1, True, True, "distributed"),        # C4
        (1100, 2, 128, "grid", 2, 2, True, False, "distributed")],      # C4
}
# n, f, dtype, inception block?, phases
DIGCN = {
    2: [(1001, 64, "float32", False, 2), (1000, 64, "bfloat16", False, 1), (1000, 64, "bfloat16", False, 2),
        # BASELINE C5: DiGCN_InceptionBlock (DiGCN_Inception_Block.py:31-47), sharded, fp32 and bf16
        (1000, 64, "float32", True, 2)],
    3: [(600, 16, "float32", False, 1)],
    4: [(1200, 64, "float32", True, 1), (1200, 64, "bfloat16", True, 2)],     # bf16, phased: fp32 partial products
    8: [(1600, 64, "bfloat16", True, 1), (1600, 64, "bfloat16", True, 2)],
}


def _suite(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // world)))   # the oracle runs in every rank: no oversubscription
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = {}
        for cfg in MAGNETIC[world]:
            out["magnetic " + str(cfg)] = _magnetic_case(rank, world, cfg)
        for cfg in DIGCN[world]:
            out["digcn " + str(cfg)] = _digcn_case(rank, world, *cfg)
        if world in (2, 4):
            for kind in ("sgcn_first", "sgcn_deep", "simpa_undirected", "simpa_directed"):
                out["signed " + kind] = _signed_case(rank, world, kind)
        if world == 4:
            out["build"] = _build_case()
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_sharded_layers_match_oracle(world):
    """Every configuration of MAGNETIC[world] / DIGCN[world] (incl. BASELINE C4 and C5 in sharded form) on `world`
    ranks; world 4 also checks the distributed operator build against the rows of the global build."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_suite, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank, out in ret.items():
        for name, v in out.items():
            if name.startswith("magnetic"):
                assert v[0] <= 1e-5, (rank, name, v)      # outputs / dX: |d| <= 1e-5 (1 + |want|)
                assert v[1] <= 1e-5, (rank, name, v)      # dW / db: max-norm (row reductions)
                # the layer's count of operator entries = the oracle's (both diagonal sets folded into one per node)
                assert v[2] == v[3], (rank, name, v)
            elif name.startswith("digcn"):
                assert v <= 1.0, (rank, name, v)
            elif name.startswith("signed"):
                assert v[0] <= 1e-5 and v[1] <= 1e-5, (rank, name, v)      # elements (mixed bar) / row reductions (max norm)
            else:
                assert v is True, (rank, name)


def _signed_case(rank, world, kind):
    """ShardedSGCNConv / ShardedSIMPA (BASELINE config 3's layers in sharded form; round 4) with the HIP kernels, against the
    un-sharded oracle layer: (worst element error |d| / (1 + |want|) over output and input gradients, worst max-norm error
    over the all-reduced parameter gradients)."""
    from oracle import ref_layers as R
    from pytorch_geometric_signed_directed_amd.parallel import ShardedSGCNConv, ShardedSIMPA, all_gather_rows
    dev = torch.device("cuda:0")
    n, f = 900, 64
    g = torch.Generator().manual_seed(31)
    pos, neg = torch.randint(0, n, (2, 6 * n), generator=g), torch.randint(0, n - 7, (2, 14 * n), generator=g)
    w_p, w_n = torch.rand(pos.size(1), generator=g) + 0.5, torch.rand(neg.size(1), generator=g) + 0.5
    torch.manual_seed(19)
    if kind.startswith("sgcn"):
        first = kind == "sgcn_first"
        layer = ShardedSGCNConv(f if first else f // 2, f // 2, first, n, pos.to(dev), neg.to(dev), device=dev)
        xs = [torch.randn(n, f, generator=g)]
    else:
        directed = kind == "simpa_directed"
        layer = ShardedSIMPA(2, 0.5, n, pos.to(dev), w_p.to(dev), neg.to(dev), w_n.to(dev), directed, device=dev)
        xs = [torch.randn(n, f, generator=g) for _ in range(4 if directed else 2)]
    with torch.no_grad():
        for prm in layer.parameters():
            prm.uniform_(0.5, 1.5)
            dist.broadcast(prm.data, 0)
    plan = layer.plan
    local = [layer.shard_rows(x.to(dev)).requires_grad_() for x in xs]
    out = layer(*local)
    go = torch.randn(n, out.size(1), generator=g)
    (out * layer.shard_rows(go.to(dev))).sum().backward()
    got = [plan.unshard_rows(all_gather_rows(t.detach())).cpu() for t in [out] + [a.grad for a in local]]
    ref_in = [x.clone().requires_grad_() for x in xs]
    sd = {k: v.detach().cpu().clone().requires_grad_() for k, v in layer.named_parameters()}
    if kind.startswith("sgcn"):
        want = R.sgcn_conv(ref_in[0], pos, neg, (sd["lin_b.weight"], sd["lin_b.bias"]), (sd["lin_u.weight"], sd["lin_u.bias"]),
                           kind == "sgcn_first", layer.in_dim)
    else:
        want = R.simpa(pos, w_p, neg, w_n, ref_in[0], ref_in[1], sd, 2, 0.5, kind == "simpa_directed", *ref_in[2:])
    (want * go).sum().backward()
    elem = max(float(((a - b).abs() / (1 + b.abs())).max()) for a, b in zip(got, [want.detach()] + [x.grad for x in ref_in]))
    red = max(float((prm.grad.cpu() - sd[k].grad).abs().max()) / max(1.0, float(sd[k].grad.abs().max()))
              for k, prm in layer.named_parameters())
    return elem, red


def _digcn_case(rank, world, n, f, dtype_name, block, phases):
    from oracle import ref_layers as R
    from pytorch_geometric_signed_directed_amd.parallel import (ShardedDiGCNConv, ShardedDiGCNInceptionBlock,
                                                                all_gather_rows)
    dev = torch.device("cuda:0")
    dtype = getattr(torch, dtype_name)
    g = torch.Generator().manual_seed(5)
    e = 12 * n
    ei, ei2 = torch.randint(0, n, (2, e), generator=g), torch.randint(0, n, (2, e), generator=g)
    w, w2 = torch.rand(e, generator=g) / 8, torch.rand(e, generator=g) / 8
    x = torch.randn(n, f, generator=g)
    go = torch.randn(n, f, generator=g)
    torch.manual_seed(13)
    if block:
        layer = ShardedDiGCNInceptionBlock(f, f, n, ei.to(dev), w.to(dev), ei2.to(dev), w2.to(dev), device=dev,
                                           phases=phases)
    else:
        layer = ShardedDiGCNConv(f, f, n, ei.to(dev), w.to(dev), device=dev, phases=phases)
    with torch.no_grad():
        for prm in layer.parameters():
            prm.uniform_(-0.5, 0.5)
            dist.broadcast(prm.data, 0)
    sd = {k: v.detach().cpu().clone() for k, v in layer.named_parameters()}
    layer.to(dtype)
    a = layer.shard_rows(x.to(dev)).to(dtype).requires_grad_()
    outs = layer(a)
    outs = outs if block else (outs,)
    sum(((k + 1.0) * o.float() * layer.shard_rows(go.to(dev))).sum() for k, o in enumerate(outs)).backward()
    got = [layer.plan.unshard_rows(all_gather_rows(t.detach().float().contiguous())).cpu() for t in outs + (a.grad,)]
    # oracle: fp32 arithmetic on inputs and parameters rounded to the storage dtype
    rnd = (lambda t: t.to(dtype).float())
    xo = rnd(x).requires_grad_()
    p = {k: rnd(v).requires_grad_() for k, v in sd.items()}
    if block:
        want = (xo @ p["ln.weight"].t() + p["ln.bias"], R.digcn_conv(xo, ei, w, p["conv1.weight"], p["conv1.bias"]),
                R.digcn_conv(xo, ei2, w2, p["conv2.weight"], p["conv2.bias"]))
    else:
        want = (R.digcn_conv(xo, ei, w, p["weight"], p["bias"]),)
    sum(((k + 1.0) * o * go).sum() for k, o in enumerate(want)).backward()
    pairs = list(zip(got, [t.detach() for t in want] + [xo.grad]))
    pairs += [(prm.grad.float().cpu(), p[k].grad) for k, prm in layer.named_parameters()]
    # fp32: the 1e-5 bar.  bf16: every stored value (projection, product, gradient) is rounded to 8 bits of
    # mantissa: relative 2^-8 of the quantity's scale per rounding, two roundings on the way to an output
    # (x W, then S^T (x W)) and three to a gradient -- bound 3 * 2^-8 of the max norm, against 3e-2 last round
    tol = 1e-5 if dtype is torch.float32 else 3 * 2.0 ** -8
    return _errs(pairs)[1] / tol


def _build_case():
    """The rows a rank assembles from the edges incident to them (degrees all-gathered) are the rows of the
    operator built whole -- same order, same values -- in both layouts."""
    from pytorch_geometric_signed_directed_amd.parallel import ShardedMagNetConv
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    n = 2000
    ei = torch.randint(0, n, (2, 30000), generator=g).to(dev)
    w = (torch.rand(30000, generator=g) + 0.5).to(dev)
    ok = True
    for layout, signed in (("rows", False), ("grid", True)):
        kw = dict(device=dev, layout=layout, signed=signed, phases=2, return_chunks=2)
        a = ShardedMagNetConv(32, 32, 1, 0.25, n, ei, w, build="distributed", **kw)
        b = ShardedMagNetConv(32, 32, 1, 0.25, n, ei, w, build="global", **kw)
        for (ca, va), (cb, vb) in zip(a.op_fwd.blocks + a.op_bwd.blocks, b.op_fwd.blocks + b.op_bwd.blocks):
            ok = ok and torch.equal(ca.rowptr, cb.rowptr) and torch.equal(ca.col, cb.col)
            ok = ok and all(torch.allclose(x, y, rtol=0, atol=1e-7) for x, y in zip(va, vb))
        ok = ok and a.global_nnz == b.global_nnz
    return bool(ok)


# ------------------------------------------------------------------ randomised sharded layers
FUZZ_ROUNDS = int(os.environ.get("PYGSD_FUZZ_ROUNDS", "4"))
FUZZ_SEED = int(os.environ.get("PYGSD_FUZZ_SEED", "1000"))


def _fuzz_magnetic_round(rank, world, seed):
    """One random sharded MagNetConv / MSConv: every rank draws the SAME case from the seed (graph with a skewed part, hub
    rows, node counts down to fewer nodes than ranks, widths that do and do not split into column slices, layout, pipeline
    depth, operator build), runs its share, and rank 0 compares the gathered rows with the un-sharded oracle in fp32 and
    float64: (description, {name: (error vs float64, the fp32 reference sequence's error vs float64)})."""
    import numpy as np
    from oracle import ref_layers as R
    from pytorch_geometric_signed_directed_amd.parallel import ShardedMagNetConv, all_gather_rows
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 2 * world)) if rng.random() < 0.15 else int(rng.integers(world, 2500))
    e = int(n * float(rng.choice([0.0, 1.0, 6.0, 20.0])))
    src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    style = int(rng.integers(0, 4))
    if e and style == 1:                                   # a third of the entries among an eighth of the nodes: uneven ranges
        k = e // 3
        src[:k], dst[:k] = rng.integers(0, max(n // 8, 1), k), rng.integers(0, max(n // 8, 1), k)
    elif e and style == 2:                                 # one hub row and one hub column
        k = int(rng.integers(1, e + 1))
        dst[:k] = rng.integers(0, n)
        src[e - min(k, e // 2):] = rng.integers(0, n)
    elif e and style == 3 and n > 2:                       # a tail of untouched nodes: ranks whose rows hold only the diagonal
        hi = int(rng.integers(1, n))
        src, dst = src % hi, dst % hi
    ei = torch.from_numpy(np.stack([src, dst]).astype(np.int64))
    signed = bool(rng.random() < 0.4)
    w = None
    if rng.random() < 0.75:
        w = torch.from_numpy((rng.random(e) + 0.25).astype(np.float32))
        if signed:
            w = w * torch.from_numpy(rng.choice([-1.0, 1.0], e).astype(np.float32))
    pool = [4, 6, 8, 16, 20, 32, 64, 72, 128]
    f_in, f_out, k = int(rng.choice(pool)), int(rng.choice(pool)), int(rng.integers(1, 4))
    q = float(rng.choice([0.25, 0.0, 0.1]))
    cols = [c for c in (2, 4, 8) if world % c == 0 and f_in % (4 * c) == 0]
    layout, grid_cols = "rows", None
    pick = rng.random()
    if pick < 0.3:
        layout = "auto"
    elif pick < 0.65 and cols:
        layout, grid_cols = "grid", int(rng.choice(cols))
    phases, chunks = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    build, balance, bias = str(rng.choice(["distributed", "global"])), bool(rng.random() < 0.7), bool(rng.random() < 0.7)
    what = (f"seed={seed} n={n} e={e} style={style} {f_in}->{f_out} K={k} q={q} signed={signed} w={w is not None} layout={layout}/"
            f"{grid_cols} phases={phases} chunks={chunks} build={build} balance={balance} bias={bias}")
    xr, xi = (torch.from_numpy(rng.standard_normal((n, f_in)).astype(np.float32)) for _ in range(2))
    gr, gi = (torch.from_numpy(rng.standard_normal((n, f_out)).astype(np.float32)) for _ in range(2))
    weight = torch.from_numpy((rng.standard_normal((k + 1, f_in, f_out)) * 0.3).astype(np.float32))
    bvec = torch.from_numpy(rng.standard_normal(f_out).astype(np.float32)) if bias else None
    layer = ShardedMagNetConv(f_in, f_out, k, q, n, ei.to(dev), None if w is None else w.to(dev), device=dev, layout=layout,
                              grid_cols=grid_cols, signed=signed, phases=phases, return_chunks=chunks, build=build,
                              balance=balance, bias=bias)
    with torch.no_grad():
        layer.weight.copy_(weight)
        if bias:
            layer.bias.copy_(bvec)
    a = layer.shard_rows(xr.to(dev)).requires_grad_()
    b = layer.shard_rows(xi.to(dev)).requires_grad_()
    o_r, o_i = layer(a, b)
    ((o_r * layer.shard_rows(gr.to(dev))).sum() + (o_i * layer.shard_rows(gi.to(dev))).sum()).backward()
    got = [layer.plan.unshard_rows(all_gather_rows(t.detach().contiguous())).cpu() for t in (o_r, o_i, a.grad, b.grad)]
    got += [layer.weight.grad.cpu()] + ([layer.bias.grad.cpu()] if bias else [])
    if rank:
        return what, {}
    refs = []
    for dtype in (torch.float32, torch.float64):
        leaf = lambda t: t.detach().clone().to(dtype).requires_grad_()      # noqa: E731
        c, d, wt = leaf(xr), leaf(xi), leaf(weight)
        bs = None if bvec is None else leaf(bvec)
        op = R.magnet_operator(ei, None if w is None else w.to(dtype), n, q, "sym", 2.0, signed=signed, absolute_degree=True,
                               dtype=dtype)
        w_r, w_i = R.magnet_conv(c, d, op, wt, bs, duplicate=False)
        ((w_r * gr.to(dtype)).sum() + (w_i * gi.to(dtype)).sum()).backward()
        refs.append([w_r.detach(), w_i.detach(), c.grad, d.grad, wt.grad] + ([bs.grad] if bias else []))
    names = ["out_real", "out_imag", "d_x_real", "d_x_imag", "d_weight", "d_bias"]
    res = {}
    for i, (g_, r32, r64) in enumerate(zip(got, *refs)):
        pair = [(g_, r64)], [(r32, r64)]
        pick_err = 1 if names[i] in ("d_weight", "d_bias") else 0        # row reductions: max norm
        res[names[i]] = (_errs(pair[0])[pick_err], _errs(pair[1])[pick_err])
    return what, res


def _fuzz_other_round(rank, world, seed):
    """One random ShardedSGCNConv / ShardedSIMPA / ShardedDiGCNConv (fp32) case, as _fuzz_magnetic_round: the graph may leave
    ranks without a single positive or negative entry, widths 4 ... 128, SIMPA hops 1 ... 3, DiGCN phases 1 ... 3."""
    import numpy as np
    from oracle import ref_layers as R
    from pytorch_geometric_signed_directed_amd.parallel import ShardedDiGCNConv, ShardedSGCNConv, ShardedSIMPA, all_gather_rows
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 2 * world)) if rng.random() < 0.1 else int(rng.integers(world, 2000))

    def edges(density):
        e = int(n * density)
        a = rng.integers(0, n, (2, e))
        if e and rng.random() < 0.3 and n > 2:            # everything among the first nodes: ranks without an entry
            a = a % int(rng.integers(1, n))
        if e and rng.random() < 0.2:
            a[1, :max(1, e // 3)] = rng.integers(0, n)    # a hub row
        return torch.from_numpy(a.astype(np.int64))

    pos, neg = edges(float(rng.choice([0.0, 2.0, 8.0]))), edges(float(rng.choice([0.0, 1.0, 10.0])))
    f32 = lambda *shape: torch.from_numpy(rng.standard_normal(shape).astype(np.float32))      # noqa: E731
    weights = lambda e: torch.from_numpy((rng.random(e) + 0.25).astype(np.float32))           # noqa: E731
    kind = str(rng.choice(["sgcn", "simpa", "digcn"]))
    pool = [4, 8, 12, 16, 32, 64, 128]
    if kind == "sgcn":
        first = bool(rng.random() < 0.5)
        out_dim = int(rng.choice(pool[:5]))
        in_dim = int(rng.choice([p_ for p_ in pool if p_ >= out_dim]))
        bias = bool(rng.random() < 0.7)
        layer = ShardedSGCNConv(in_dim, out_dim, first, n, pos.to(dev), neg.to(dev), bias=bias, device=dev)
        xs = [f32(n, in_dim if first else 2 * in_dim)]
        what = f"sgcn {in_dim}->{out_dim} first={first} bias={bias}"
    elif kind == "simpa":
        hop, fill, directed = int(rng.integers(1, 4)), float(rng.choice([0.5, 1.0, 0.2])), bool(rng.random() < 0.5)
        w_p, w_n = weights(pos.size(1)), weights(neg.size(1))
        f = int(rng.choice(pool))
        layer = ShardedSIMPA(hop, fill, n, pos.to(dev), w_p.to(dev), neg.to(dev), w_n.to(dev), directed, device=dev)
        xs = [f32(n, f) for _ in range(4 if directed else 2)]
        what = f"simpa f={f} hop={hop} fill={fill} directed={directed}"
    else:
        f_in, f_out, phases = int(rng.choice(pool)), int(rng.choice(pool)), int(rng.integers(1, 4))
        w_p = (f32(pos.size(1)) * 0.3)
        layer = ShardedDiGCNConv(f_in, f_out, n, pos.to(dev), w_p.to(dev), device=dev, phases=phases)
        xs = [f32(n, f_in)]
        what = f"digcn {f_in}->{f_out} phases={phases}"
    what = f"seed={seed} n={n} e+={pos.size(1)} e-={neg.size(1)} " + what
    values = {k: f32(*v.shape) * 0.4 + (0.5 if kind == "simpa" else 0.0) for k, v in layer.named_parameters()}
    with torch.no_grad():
        for k, prm in layer.named_parameters():
            prm.copy_(values[k])
    local = [layer.shard_rows(x.to(dev)).requires_grad_() for x in xs]
    out = layer(*local)
    go = f32(n, out.size(1))
    (out * layer.shard_rows(go.to(dev))).sum().backward()
    got = [layer.plan.unshard_rows(all_gather_rows(t.detach().contiguous())).cpu() for t in [out] + [a.grad for a in local]]
    got += [prm.grad.cpu() if prm.grad is not None else torch.zeros_like(prm).cpu() for _, prm in layer.named_parameters()]
    if rank:
        return what, {}
    refs = []
    for dtype in (torch.float32, torch.float64):
        leaf = lambda t: t.detach().clone().to(dtype).requires_grad_()      # noqa: E731
        ri = [leaf(x) for x in xs]
        sd = {k: leaf(v) for k, v in values.items()}
        if kind == "sgcn":
            want = R.sgcn_conv(ri[0], pos, neg, (sd["lin_b.weight"], sd.get("lin_b.bias")), (sd["lin_u.weight"], sd.get("lin_u.bias")),
                               first, in_dim)
        elif kind == "simpa":
            want = R.simpa(pos, w_p.to(dtype), neg, w_n.to(dtype), ri[0], ri[1], sd, hop, fill, directed, *ri[2:])
        else:
            want = R.digcn_conv(ri[0], pos, w_p.to(dtype), sd["weight"], sd.get("bias"))
        (want * go.to(dtype)).sum().backward()
        zero = lambda t: t.grad if t.grad is not None else torch.zeros_like(t)      # noqa: E731
        refs.append([want.detach()] + [zero(x) for x in ri] + [zero(sd[k]) for k, _ in layer.named_parameters()])
    names = ["out"] + [f"d_x{i}" for i in range(len(xs))] + ["d_" + k for k, _ in layer.named_parameters()]
    res = {}
    for i, (g_, r32, r64) in enumerate(zip(got, *refs)):
        pick_err = 1 if i > len(xs) else 0                                   # parameters: row reductions, max norm
        res[names[i]] = (_errs([(g_, r64)])[pick_err], _errs([(r32, r64)])[pick_err])
    return what, res


def _fuzz_twice_round(rank, world, seed):
    """One sharded layer applied to two inputs before ONE backward of the summed objective, against two separate forward /
    backward passes: the engine's send / receive / return buffers are reused by every call, so nothing the first forward
    saved for its backward may live in them.  Input-gradient shards bit for bit, parameter gradients to 1e-5 (max norm);
    (description, worst input-gradient difference over all ranks, worst parameter-gradient error over all ranks)."""
    import numpy as np
    from pytorch_geometric_signed_directed_amd.parallel import ShardedDiGCNConv, ShardedMagNetConv, ShardedSGCNConv, ShardedSIMPA
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    n = int(rng.integers(world, 1500))
    e = int(n * float(rng.choice([1.0, 6.0, 15.0])))
    ei = torch.from_numpy(rng.integers(0, n, (2, e)).astype(np.int64)).to(dev)
    ei2 = torch.from_numpy(rng.integers(0, n, (2, max(1, e // 2))).astype(np.int64)).to(dev)
    w = torch.from_numpy((rng.random(e) + 0.25).astype(np.float32)).to(dev)
    w2 = torch.from_numpy((rng.random(ei2.size(1)) + 0.25).astype(np.float32)).to(dev)
    f = int(rng.choice([8, 16, 32, 64]))
    kind = str(rng.choice(["magnet_rows", "magnet_grid", "magnet_k1", "digcn", "sgcn", "simpa"]))
    phases, chunks = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    torch.manual_seed(seed)
    if kind.startswith("magnet"):
        cols = [c for c in (2, 4, 8) if world % c == 0 and f % (4 * c) == 0]
        grid = kind == "magnet_grid" and cols
        layer = ShardedMagNetConv(f, f, 1 if kind == "magnet_k1" else 2, 0.25, n, ei, w, device=dev, layout="grid" if grid else "rows",
                                  grid_cols=int(rng.choice(cols)) if grid else None, phases=phases, return_chunks=chunks)
        n_in, call = 2, (lambda a, b: layer(a, b))
    elif kind == "digcn":
        layer = ShardedDiGCNConv(f, f, n, ei, w * 0.2, device=dev, phases=phases)
        n_in, call = 1, (lambda a, b: layer(a))
    elif kind == "sgcn":
        layer = ShardedSGCNConv(f, f // 2, True, n, ei, ei2, device=dev)
        n_in, call = 1, (lambda a, b: layer(a))
    else:
        layer = ShardedSIMPA(2, 0.5, n, ei, w, ei2, w2, False, device=dev)
        n_in, call = 2, (lambda a, b: layer(a, b))
    with torch.no_grad():
        for prm in layer.parameters():
            prm.uniform_(-0.5, 0.5)
            dist.broadcast(prm.data, 0)
    xs = [layer.shard_rows(torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32)).to(dev)) for _ in range(4)]

    def objective(out, salt):
        out = out if isinstance(out, (tuple, list)) else (out,)
        return sum((o * torch.sin(torch.arange(o.numel(), device=dev, dtype=torch.float32).view_as(o) * (0.37 + salt) + k)).sum()
                   for k, o in enumerate(out))

    params = list(layer.parameters())

    def run(joint):
        leaves = [x.clone().requires_grad_() for x in xs]
        for prm in params:
            prm.grad = None
        if joint:
            (objective(call(leaves[0], leaves[1]), 0.0) + objective(call(leaves[2], leaves[3]), 0.5)).backward()
        else:
            objective(call(leaves[0], leaves[1]), 0.0).backward()
            objective(call(leaves[2], leaves[3]), 0.5).backward()
        used = [0, 1, 2, 3] if n_in == 2 else [0, 2]
        return [leaves[i].grad.clone() for i in used], [prm.grad.clone() for prm in params]

    gj, pj = run(True)
    ga, pa = run(False)
    worst_x = max(float((a - b).abs().max()) if a.numel() else 0.0 for a, b in zip(gj, ga))
    worst_p = max(float((a - b).abs().max()) / max(1.0, float(b.abs().max())) for a, b in zip(pj, pa))
    t = torch.tensor([worst_x, worst_p], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return f"seed={seed} {kind} n={n} e={e} f={f} phases={phases} chunks={chunks}", float(t[0]), float(t[1])


def _fuzz_input_cache_round(rank, world, seed):
    """The opt-in input-exchange memo (`cache_input_exchange=True`: while x_real / x_imag are the same tensors at the same
    in-place version, the forward's inbound exchange is not repeated) under a random history of steps -- same tensors,
    in-place edits through torch, fresh tensors, a backward in between: every output within 1e-5 (1 + |.|) of a twin layer's
    without the memo.  (description, worst such difference over all steps and ranks)."""
    import numpy as np
    from pytorch_geometric_signed_directed_amd.parallel import ShardedMagNetConv
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    n = int(rng.integers(world, 1200))
    e = int(n * float(rng.choice([2.0, 10.0])))
    ei = torch.from_numpy(rng.integers(0, n, (2, e)).astype(np.int64)).to(dev)
    f = int(rng.choice([8, 16, 32]))
    cols = [c for c in (2, 4, 8) if world % c == 0 and f % (4 * c) == 0]
    grid = bool(cols) and rng.random() < 0.5
    kw = dict(device=dev, layout="grid" if grid else "rows", grid_cols=int(rng.choice(cols)) if grid else None,
              phases=int(rng.integers(1, 3)), return_chunks=int(rng.integers(1, 3)))
    torch.manual_seed(seed)
    cached = ShardedMagNetConv(f, f, int(rng.integers(1, 3)), 0.25, n, ei, None, cache_input_exchange=True, **kw)
    plain = ShardedMagNetConv(f, f, cached.weight.size(0) - 1, 0.25, n, ei, None, cache_input_exchange=False, **kw)
    with torch.no_grad():
        for a, b in zip(cached.parameters(), plain.parameters()):
            dist.broadcast(a.data, 0)
            b.copy_(a)
    fresh = lambda: cached.shard_rows(torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32)).to(dev))   # noqa: E731
    xr, xi = fresh(), fresh()
    differ, history = 0.0, []
    for _ in range(int(rng.integers(4, 9))):
        act = str(rng.choice(["same", "same", "edit_real", "edit_imag", "new_real", "new_both", "train_step"]))
        history.append(act)
        if act == "edit_real" and xr.numel():
            xr[int(rng.integers(0, xr.size(0))), int(rng.integers(0, f))] += 1.0
        elif act == "edit_imag" and xi.numel():
            xi.mul_(1.5)
        elif act == "new_real":
            xr = fresh()
        elif act == "new_both":
            xr, xi = fresh(), fresh()
        if act == "train_step":
            outs = []
            for layer in (cached, plain):
                a, b = xr.clone().requires_grad_(), xi.clone().requires_grad_()
                o = layer(a, b)
                (o[0].sum() + o[1].sum()).backward()
                outs.append((o[0].detach(), o[1].detach(), a.grad, b.grad))
        else:
            with torch.no_grad():
                outs = [layer(xr, xi) for layer in (cached, plain)]
        # (a remembered exchange lets the product run over all columns at once instead of phase by phase: another order of
        #  the same fp32 sums -- rounding apart, never the O(1) of a stale input: the edits add 1.0 / scale by 1.5)
        differ = max([differ] + [float(((p_ - q_).abs() / (1.0 + q_.abs())).max()) for p_, q_ in zip(*outs) if p_.numel()])
    t = torch.tensor([float(differ)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return f"seed={seed} n={n} e={e} f={f} {kw['layout']} history={history}", float(t[0])


def _fuzz_suite(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // world)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = []
        for r in range(FUZZ_ROUNDS):
            out.append(_fuzz_magnetic_round(rank, world, FUZZ_SEED + 7919 * world + r))
            out.append(_fuzz_other_round(rank, world, FUZZ_SEED + 104729 * world + r))
            out.append(("twice",) + _fuzz_twice_round(rank, world, FUZZ_SEED + 15485863 * world + r))
            out.append(("input_cache",) + _fuzz_input_cache_round(rank, world, FUZZ_SEED + 32452843 * world + r))
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_fuzz_sharded_layers(world):
    """Random sharded MagNetConv / MSConv and SGCNConv / SIMPA / DiGCNConv cases (see _fuzz_magnetic_round, _fuzz_other_round)
    on `world` ranks sharing the test box's GPU; every
    output and gradient within max(1e-5, 3 x the fp32 reference sequence's own error) of float64 -- the bar, and the reason
    for the 3, of tests/test_gpu_fuzz.py.  PYGSD_FUZZ_ROUNDS / PYGSD_FUZZ_SEED as there."""
    ret = mp.Manager().dict()
    mp.spawn(_fuzz_suite, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    bad, gross, checks = [], [], 0
    twice = [r for r in ret[0] if r[0] == "twice"]
    stale = [f"{what}: input gradients differ by {dx:.3e}, parameter gradients by {dp:.3e}" for _, what, dx, dp in twice
             if dx != 0.0 or not dp <= 1e-5]
    assert not stale, "two forwards before one backward:\n" + "\n".join(stale[:20])
    stale = [f"{what}: differs by {k:.3e}" for tag, what, k in (r for r in ret[0] if r[0] == "input_cache") if not k <= 1e-5]
    assert not stale, "input-exchange memo served a stale exchange:\n" + "\n".join(stale[:20])
    for what, res in (r for r in ret[0] if r[0] not in ("twice", "input_cache")):
        for name, (mine, theirs) in res.items():
            checks += 1
            if not mine <= max(1e-5, 3.0 * theirs):
                bad.append(f"{name}: {mine:.3e} > max(1e-5, 3 x {theirs:.3e}) [{what}]")
                if not mine <= 50.0 * max(1e-5, 3.0 * theirs):
                    gross.append(bad[-1])
    if bad and os.environ.get("PYGSD_FUZZ_LOG"):
        with open(os.environ["PYGSD_FUZZ_LOG"], "a") as fh:
            fh.write("\n".join(f"sharded world={world} " + b for b in bad) + "\n")
    # (outliers: logged; a failure is a gross one or more than 1 in 200 checks -- tests/test_gpu_fuzz.py says why)
    assert not gross, "\n".join(gross[:20])
    assert len(bad) <= max(1, checks // 200), f"{len(bad)} of {checks} checks:\n" + "\n".join(bad[:20])


@pytest.mark.parametrize("gpus", [2, 4, 8])
def test_bench_self_launches_its_ranks(gpus):
    """`python bench.py --gpus N` as typed (no torchrun): bench.py re-executes itself under torch.distributed.run,
    one rank per GPU; here all ranks share the test GPU over gloo (PYGSD_BENCH_SHARE_GPU hook).  One JSON line on
    stdout with the multi-GPU fields."""
    env = dict(os.environ, PYGSD_BENCH_SHARE_GPU="1", PYGSD_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2",
                          "--warmup", "1", "--nodes", "20000", "--edges", "300000", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == gpus and rec["value"] > 0 and rec["scaling"] == "strong"
    assert rec["config"]["nodes"] == 20000 and "exchange" in rec and rec["exchange"]["propagates_per_step"] == 2
    for key in ("product_ms", "exposed_exchange_ms", "exchange_alone_ms", "collectives_alone", "fallback"):
        assert key in rec["exchange"], rec["exchange"]
    assert rec["exchange"]["fallback"] is None
    # who ran where (round 6): the line proves its N ranks by itself -- one identity record per rank (here all on the test GPU)
    assert rec["exchange"]["world_size"] == gpus and len(rec["exchange"]["devices"]["ranks"]) == gpus, rec["exchange"]
    assert sorted(r["rank"] for r in rec["exchange"]["devices"]["ranks"]) == list(range(gpus))
    assert all(r["uuid"] and r["name"] for r in rec["exchange"]["devices"]["ranks"]), rec["exchange"]["devices"]
    # the un-timed parity guard: sampled rows of the sharded run against the un-sharded HIP layer, in the line
    assert rec["parity"]["ok"] and rec["parity"]["rows"] == 1024 and rec["parity"]["max_err_rows"] <= 1e-5, rec["parity"]



def test_rccl_path_with_one_rank():
    """The RCCL ("nccl") code path of the exchanges -- asynchronous stacked all_gather_into_tensor, equal-split
    all_to_all_single, all_reduce, object broadcast -- and of bench.py's sharded mode, with the one rank a 1-GPU box
    allows (RCCL refuses two ranks per device).  Runs in subprocesses: a process group per process."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PYGSD_DIST_BACKEND", None)
    env.pop("PYGSD_BENCH_SHARE_GPU", None)
    code = """
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from pytorch_geometric_signed_directed_amd.parallel import DistExchange, all_gather_rows
ex = DistExchange()
assert not ex._staged and ex.world_size == 1
x = torch.arange(24, dtype=torch.float32, device="cuda").view(4, 6)
out = torch.empty(1, 4, 6, device="cuda")
ex.all_gather(out, x).wait()
assert torch.equal(out[0], x)
inp = x.view(1, 4, 6).clone()
back = torch.empty(3, 1, 4, 6, device="cuda")
works = [ex.all_to_all(back[r], inp * (r + 1)) for r in range(3)]
for w in works:
    w.wait()
assert all(torch.equal(back[r, 0], x * (r + 1)) for r in range(3))
t = torch.ones(5, dtype=torch.float64, device="cuda")
assert float(ex.all_reduce(t).sum()) == 5.0
assert ex.broadcast_list([0, 3, 7]) == [0, 3, 7]
assert torch.equal(all_gather_rows(x), x)
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
"""
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stderr[-2000:]
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    # the WHOLE pipelined schedule under RCCL at the north-star size (1M nodes / 20M edges, h = 64): with one column
    # slice forced (--layout grid --grid-cols 1) the single rank runs the grid's asynchronous all-to-all in two phases,
    # the row-chunked all-to-all back, the merge and the asynchronous dW / db all-reduce on the process group's stream,
    # and bench.py's parity guard compares sampled rows and dW / db with the un-sharded layer
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-sharded", "--layout", "grid",
                          "--grid-cols", "1", "--phases", "2", "--return-chunks", "2", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    ex = rec["exchange"]
    assert rec["n_gpus"] == 1 and rec["config"]["nodes"] == 1000000 and rec["value"] > 0
    assert (ex["layout"], ex["p_r"], ex["p_c"], ex["phases"], ex["return_chunks"]) == ("grid", 1, 1, 2, 2), ex
    assert ex["backend"] == "nccl" and ex["fallback"] is None and not ex["blocking_collectives"]
    assert ex["TORCH_NCCL_AVOID_RECORD_STREAMS"] == "1"
    assert len(ex["collectives_alone"]["inbound_ms_per_phase"]) == 2 and len(ex["collectives_alone"]["return_ms_per_chunk"]) == 2
    assert rec["parity"]["ok"] and rec["parity"]["max_err_rows"] <= 1e-5, rec["parity"]
    # the row layout (stacked all-gather) the same way, small
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-sharded", "--steps", "2",
                          "--warmup", "1", "--nodes", "20000", "--edges", "300000", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert rec["n_gpus"] == 1 and rec["exchange"]["layout"] == "rows" and rec["value"] > 0 and rec["parity"]["ok"]
