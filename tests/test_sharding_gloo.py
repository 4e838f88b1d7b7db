"""CPU, gloo, 2 .. 8 ranks: the node-range sharding of pytorch_geometric_signed_directed_amd.parallel end to end --
equal-work ownership plan, padded ids, column phases, interleaved row blocks, row-chunked returns, packing,
both exchanges, merges, the Chebyshev recurrence / its adjoint over exchanged blocks, parameter all-reduce -- with
the two product kernels swapped for torch restatements (tests/sharding_cpu.py) and the operator rows taken
from the oracle.  Every rank must reproduce its rows of the UN-SHARDED oracle layer, outputs and gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_layers as R
from pytorch_geometric_signed_directed_amd.parallel import (PropagateEngine, ShardPlan, balanced_bounds, choose_cols,
                                                            degree_cost, split_phases, take_rows)
from pytorch_geometric_signed_directed_amd.sparse import CSR
import sharding_cpu as C


# ------------------------------------------------------------------------------------------------
# single-process: plan, CSR surgery, engine geometry
# ------------------------------------------------------------------------------------------------
def test_plan_partitions_every_node_once_and_pads():
    for n, w, align in [(10, 2, 1), (11, 3, 4), (7, 8, 2), (1000, 8, 8), (1, 2, 1)]:
        seen = []
        for r in range(w):
            p = ShardPlan(n, w, r, align=align)
            assert p.n_pad % align == 0 and p.n_total == w * p.n_pad and p.n_pad >= max(p.sizes)
            seen += list(range(p.lo, p.hi))
            x = torch.arange(n, dtype=torch.float32).unsqueeze(1)
            s = p.shard_rows(x)
            assert s.shape == (p.n_pad, 1)
            assert s[:p.n_local, 0].tolist() == list(range(p.lo, p.hi)) and float(s[p.n_local:].abs().sum()) == 0
        assert seen == list(range(n))


def test_padded_ids_round_trip_with_uneven_and_empty_ranges():
    p = ShardPlan(10, 4, 2, bounds=[0, 5, 5, 6, 10], align=3)
    assert p.sizes == [5, 0, 1, 4] and p.n_pad == 6 and (p.lo, p.hi, p.n_local, p.pad_lo) == (5, 6, 1, 12)
    ids = torch.arange(10)
    assert p.to_padded(ids).tolist() == [0, 1, 2, 3, 4, 12, 18, 19, 20, 21]
    x = torch.randn(10, 3)
    gathered = torch.cat([ShardPlan(10, 4, r, bounds=p.bounds, align=3).shard_rows(x) for r in range(4)])
    assert torch.equal(p.unshard_rows(gathered), x)
    with pytest.raises(ValueError):
        ShardPlan(10, 2, 0, bounds=[0, 11, 10])


def test_balanced_bounds_equalise_work_not_size():
    """A graph whose first nodes carry almost all the edges (an un-permuted SBM with a dense block): equal-size
    ranges put ~all the entries on rank 0, equal-work ranges spread them (SURVEY.md 8(e) "Load balance")."""
    g = torch.Generator().manual_seed(0)
    n, world = 4000, 4
    heavy = torch.randint(0, 400, (2, 36000), generator=g)
    light = torch.randint(0, n, (2, 4000), generator=g)
    ei = torch.cat([heavy, light], dim=1)
    cost = degree_cost(ei, n)
    bounds = balanced_bounds(cost, world)
    work = [float(cost[bounds[r]:bounds[r + 1]].sum()) for r in range(world)]
    even = [float(cost[r * 1000:(r + 1) * 1000].sum()) for r in range(world)]
    assert max(work) / (sum(work) / world) < 1.05 < 3.0 < max(even) / (sum(even) / world)
    assert bounds[0] == 0 and bounds[-1] == n and sorted(bounds) == bounds
    assert balanced_bounds(torch.ones(7), 7) == list(range(8))
    assert balanced_bounds(torch.zeros(0), 3) == [0, 0, 0, 0]


def _random_csr(n_rows, n_cols, nnz, seed):
    g = torch.Generator().manual_seed(seed)
    rows, cols = torch.randint(0, n_rows, (nnz,), generator=g), torch.randint(0, n_cols, (nnz,), generator=g)
    csr = C.cpu_csr_from_coo(rows, cols, n_rows, n_cols)
    return csr, torch.randn(nnz, generator=g)


def _apply(csr, val, x):
    y = torch.zeros(csr.n_rows, x.size(1))
    C.cpu_single(csr, val, x, y, 0, csr.n_rows, 1.0, False, False)
    return y


def test_take_rows_and_split_phases_preserve_the_product():
    world, n_pad, phases = 3, 8, 4
    n = world * n_pad
    csr, val = _random_csr(n, n, 300, 1)
    x = torch.randn(n, 5, generator=torch.Generator().manual_seed(2))
    want = _apply(csr, val, x)
    rows = torch.tensor([5, 0, 23, 7, 7])
    sub, (v,) = take_rows(csr, (val,), rows)
    assert torch.equal(_apply(sub, v, x), want[rows])
    blocks = split_phases(csr, (val,), n_pad, phases, world)
    assert sum(b[0].nnz for b in blocks) == csr.nnz
    n_sub = n_pad // phases
    acc = torch.zeros_like(want)
    for c, (blk, (vc,)) in enumerate(blocks):
        # the exchange buffer of phase c: sub-range c of every rank, rank-major
        buf = x.view(world, n_pad, 5)[:, c * n_sub:(c + 1) * n_sub].reshape(world * n_sub, 5)
        assert blk.n_cols == world * n_sub
        acc += _apply(blk, vc, buf)
    assert torch.allclose(acc, want, atol=1e-5)


def test_uneven_pipeline_pieces():
    """Round 5: phases / return chunks as FRACTIONS (a short first inbound phase, a short last return chunk): the cut points,
    the spec parser, and split_phases over uneven column sub-ranges reproducing the product."""
    from pytorch_geometric_signed_directed_amd.parallel import cut_points, split_spec
    assert cut_points(12, 3) == [0, 4, 8, 12]
    assert cut_points(100, 2, (0.4, 0.6)) == [0, 40, 100]
    assert cut_points(125066, 3, (0.5, 0.36, 0.14)) == [0, 62533, 107557, 125066]
    assert cut_points(3, 3, (0.98, 0.01, 0.01)) == [0, 1, 2, 3]              # every piece keeps a row
    assert cut_points(2, 3, (1, 1, 1))[-1] == 2                               # fewer rows than pieces: some are empty
    with pytest.raises(ValueError):
        cut_points(10, 3)
    with pytest.raises(ValueError):
        cut_points(10, 2, (0.5, -0.5))
    assert split_spec(2) == (2, None) and split_spec("3") == (3, None)
    assert split_spec((0.4, 0.6)) == (2, (0.4, 0.6)) and split_spec("0.5,0.3,0.2") == (3, (0.5, 0.3, 0.2))
    assert split_spec((1.0,)) == (1, None)
    assert split_spec("2.0") == (2, None)                                     # one number is a count, however it is written
    with pytest.raises(ValueError):
        split_spec("0.5")                                                     # ... and a lone fraction is neither
    assert PropagateEngine.alignment(8, 4, (0.4, 0.6), (0.5, 0.3, 0.2)) == 2 and PropagateEngine.alignment(8, 4, 2, 2) == 4
    world, n_pad = 3, 11
    n = world * n_pad
    csr, val = _random_csr(n, n, 400, 3)
    x = torch.randn(n, 5, generator=torch.Generator().manual_seed(4))
    want = _apply(csr, val, x)
    for bounds in ([0, 3, 11], [0, 1, 2, 11], [0, 0, 11], [0, 11, 11]):
        blocks = split_phases(csr, (val,), n_pad, len(bounds) - 1, world, bounds)
        assert sum(b[0].nnz for b in blocks) == csr.nnz
        acc = torch.zeros_like(want)
        for c, (blk, (vc,)) in enumerate(blocks):
            buf = x.view(world, n_pad, 5)[:, bounds[c]:bounds[c + 1]].reshape(-1, 5)
            assert blk.n_cols == world * (bounds[c + 1] - bounds[c])
            acc += _apply(blk, vc, buf)
        assert torch.allclose(acc, want, atol=1e-5)
    with pytest.raises(ValueError):
        split_phases(csr, (val,), n_pad, 2, world, [0, 5, 10])


def test_split_phases_of_an_empty_shard():
    """A rank that multiplies no rows (or rows without entries) still takes part in every phase: empty blocks of the
    phase buffers' shape instead of a zero slice step."""
    from pytorch_geometric_signed_directed_amd.sparse import CSR
    world, n_pad, phases = 2, 8, 2
    for n_rows in (0, 3):
        csr = CSR(n_rows, world * n_pad, 0, torch.zeros(n_rows + 1, dtype=torch.int32), torch.zeros(0, dtype=torch.int32), None)
        blocks = split_phases(csr, (torch.zeros(0),), n_pad, phases, world)
        assert len(blocks) == phases
        for blk, (v,) in blocks:
            assert (blk.n_rows, blk.n_cols, blk.nnz) == (n_rows, world * n_pad // phases, 0)
            assert blk.rowptr.numel() == n_rows + 1 and int(blk.rowptr.abs().sum()) == 0 and v.numel() == 0


@pytest.mark.parametrize("world,p_c,phases,chunks", [(8, 4, 2, 2), (4, 4, 1, 3), (8, 2, 3, 1), (4, 1, 2, 1), (6, 2, 2, 2),
                                                     (8, 4, (0.4, 0.6), (0.5, 0.36, 0.14)), (4, 2, 2, (0.7, 0.3))])
def test_row_blocks_cover_every_padded_row_exactly_once(world, p_c, phases, chunks):
    align = PropagateEngine.alignment(world, p_c, phases, chunks)
    seen = torch.zeros(0, dtype=torch.long)
    for rank in range(world):
        plan = ShardPlan(1000, world, rank, align=align)
        eng = PropagateEngine(plan, type("Ex", (), {"world_size": world, "rank": rank})(), p_c, phases, chunks, C.KERNELS)
        ids = eng.block_row_ids("cpu")
        assert ids.numel() == eng.block_rows and ids.unique().numel() == ids.numel()
        if eng.j == 0:
            seen = torch.cat([seen, ids])
    assert torch.equal(seen.sort().values, torch.arange(plan.n_total))
    assert [choose_cols(w, 64) for w in (1, 2, 4, 6, 8)] == [1, 1, 4, 2, 4]
    assert choose_cols(8, 12) == 1 and choose_cols(8, 24) == 2              # 16-byte aligned slices only


# ------------------------------------------------------------------------------------------------
# multi-process
# ------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _graph(n, seed, skew):
    g = torch.Generator().manual_seed(seed)
    e = 10 * n
    ei = torch.randint(0, n, (2, e), generator=g)
    if skew:                                        # most edges among the first tenth of the nodes
        ei[:, :e // 2] = torch.randint(0, max(n // 10, 2), (2, e // 2), generator=g)
    w = torch.rand(e, generator=g) + 0.5
    return g, ei, w


def _magnetic_worker(rank, world, port, cfg, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // world)))   # the oracle runs in every rank: no oversubscription
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pytorch_geometric_signed_directed_amd.parallel import ShardedMagNetConv, all_gather_rows
        n, f, k, layout, phases, chunks, signed, skew = cfg
        g, ei, w = _graph(n, 123, skew)
        if signed:
            w = w * (torch.randint(0, 2, w.shape, generator=g) * 2 - 1)
        xr, xi = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
        gr, gi = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
        torch.manual_seed(11)
        layer = ShardedMagNetConv(f, f, k, 0.25, n, ei, w, signed=signed, layout=layout, phases=phases,
                                  return_chunks=chunks, grid_cols=2 if (layout == "grid" and world == 2) else None,
                                  kernels=C.KERNELS, operator_rows=C.oracle_operator_rows(ei, w, n, 0.25, signed=signed))
        from pytorch_geometric_signed_directed_amd.parallel import default_pipeline, split_spec
        want_phases = default_pipeline(world, layout == "grid")[0] if phases is None else phases
        assert layer.layout == layout and layer.engine.phases == split_spec(want_phases)[0]
        plan = layer.plan
        if skew:
            assert plan.sizes != [plan.sizes[0]] * world            # balanced ranges are uneven here
        with torch.no_grad():
            layer.bias.uniform_(-0.5, 0.5)
            dist.broadcast(layer.bias.data, 0)
        a, b = layer.shard_rows(xr).requires_grad_(), layer.shard_rows(xi).requires_grad_()
        o_r, o_i = layer(a, b)
        # the loss deliberately touches the pad rows: their upstream gradient must not leak into dW / db
        ((o_r * layer.shard_rows(gr)).sum() + (o_i * layer.shard_rows(gi)).sum() + 3.0 * o_r[plan.n_local:].sum()).backward()
        got = [plan.unshard_rows(all_gather_rows(t.detach())) for t in (o_r, o_i, a.grad, b.grad)]
        # un-sharded oracle (reference op sequence)
        weight, bias = layer.weight.detach().clone(), layer.bias.detach().clone()
        c, d = xr.clone().requires_grad_(), xi.clone().requires_grad_()
        wt, bs = weight.clone().requires_grad_(), bias.clone().requires_grad_()
        op = R.magnet_operator(ei, w, n, 0.25, "sym", 2.0, signed=signed)
        w_r, w_i = R.magnet_conv(c, d, op, wt, bs, duplicate=False)
        ((w_r * gr).sum() + (w_i * gi).sum()).backward()
        worst = 0.0
        for x, y in zip(got + [layer.weight.grad, layer.bias.grad], [w_r.detach(), w_i.detach(), c.grad, d.grad, wt.grad, bs.grad]):
            worst = max(worst, float((x - y).abs().max()) / max(1.0, float(y.abs().max())))
        ret[rank] = worst
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,cfg", [
    # n, f, K, layout, phases, return chunks, signed, skewed graph
    (2, (40, 8, 1, "rows", 1, 1, False, False)),
    (2, (41, 8, 2, "rows", 2, 1, False, True)),
    (3, (50, 4, 3, "rows", 3, 1, True, True)),
    (2, (40, 8, 2, "grid", 2, 2, False, False)),        # 1 x 2
    (4, (45, 16, 3, "grid", 2, 3, True, True)),         # 1 x 4, Chebyshev order 3 through the grid
    (8, (70, 16, 2, "grid", 2, 2, False, True)),        # 2 x 4: the 8-GPU configuration, interleaved row blocks
    (8, (64, 16, 1, "grid", 1, 1, True, False)),        # 2 x 4 un-pipelined
    (6, (50, 8, 2, "grid", 2, 2, False, False)),        # 3 x 2
    (8, (70, 16, 2, "grid", (0.4, 0.6), (0.5, 0.36, 0.14), False, True)),   # 2 x 4, round 5's default: uneven phases / return chunks
    (4, (45, 16, 1, "grid", (0.3, 0.3, 0.4), (0.8, 0.2), True, True)),      # 1 x 4, three uneven phases
    (3, (50, 4, 2, "rows", (0.25, 0.75), 1, False, True)),                  # rows layout, uneven phases
    (8, (70, 16, 1, "grid", None, None, False, False)),                     # whatever the defaults are
])
def test_sharded_magnetic_layer_equals_unsharded_oracle(world, cfg):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_magnetic_worker, args=(world, _free_port(), cfg, ret), nprocs=world, join=True)
    assert len(ret) == world and max(ret.values()) <= 2e-6, dict(ret)


def _stacked_worker(rank, world, port, ret):
    """Two sharded magnetic layers stacked, uneven ranges (pad rows on most ranks), non-zero biases, a loss that also
    touches the pad rows of the SECOND layer: outputs on pad rows must be zero (not the bias), and no upstream
    gradient of a pad row may reach dW / db of either layer."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pytorch_geometric_signed_directed_amd.parallel import ShardedMagNetConv, all_gather_rows
        n, f, k = 53, 8, 2
        g, ei, w = _graph(n, 77, True)
        xr, xi = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
        gr, gi = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
        torch.manual_seed(5)
        layers = [ShardedMagNetConv(f, f, k, 0.25, n, ei, w, layout="rows", phases=1, kernels=C.KERNELS,
                                    operator_rows=C.oracle_operator_rows(ei, w, n, 0.25)) for _ in range(2)]
        plan = layers[0].plan
        assert plan.n_local < plan.n_pad or rank == max(range(world), key=lambda r: plan.sizes[r])
        with torch.no_grad():
            for ly in layers:
                ly.bias.uniform_(0.5, 1.5)
                dist.broadcast(ly.bias.data, 0)
                dist.broadcast(ly.weight.data, 0)
        a, b = plan.shard_rows(xr).requires_grad_(), plan.shard_rows(xi).requires_grad_()
        h_r, h_i = layers[0](a, b)
        assert float(h_r[plan.n_local:].abs().sum()) == 0 and float(h_i[plan.n_local:].abs().sum()) == 0
        o_r, o_i = layers[1](h_r, h_i)
        assert float(o_r[plan.n_local:].abs().sum()) == 0
        ((o_r * plan.shard_rows(gr)).sum() + (o_i * plan.shard_rows(gi)).sum() + 3.0 * o_r.sum() - 2.0 * o_i.sum()
         + 5.0 * h_i[plan.n_local:].sum()).backward()
        got = [plan.unshard_rows(all_gather_rows(t.detach())) for t in (o_r, o_i, a.grad, b.grad)]
        c, d = xr.clone().requires_grad_(), xi.clone().requires_grad_()
        prm = [(ly.weight.detach().clone().requires_grad_(), ly.bias.detach().clone().requires_grad_()) for ly in layers]
        op = R.magnet_operator(ei, w, n, 0.25, "sym", 2.0)
        m_r, m_i = R.magnet_conv(c, d, op, prm[0][0], prm[0][1], duplicate=False)
        w_r, w_i = R.magnet_conv(m_r, m_i, op, prm[1][0], prm[1][1], duplicate=False)
        ((w_r * gr).sum() + (w_i * gi).sum() + 3.0 * w_r.sum() - 2.0 * w_i.sum()).backward()
        want = [w_r.detach(), w_i.detach(), c.grad, d.grad] + [t.grad for pair in prm for t in pair]
        have = got + [t for ly in layers for t in (ly.weight.grad, ly.bias.grad)]
        ret[rank] = max(float((x - y).abs().max()) / max(1.0, float(y.abs().max())) for x, y in zip(have, want))
    finally:
        dist.destroy_process_group()


def test_stacked_sharded_layers_keep_pad_rows_out_of_the_graph():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_stacked_worker, args=(3, _free_port(), ret), nprocs=3, join=True)
    assert len(ret) == 3 and max(ret.values()) <= 2e-6, dict(ret)


@pytest.mark.parametrize("world,layout,k,phases,chunks", [(4, "grid", 2, 2, 2), (8, "grid", 1, 2, 2), (3, "rows", 3, 2, 1)])
def test_ranks_as_threads_equal_the_unsharded_oracle(world, layout, k, phases, chunks):
    """parallel.ThreadExchange: the ranks of a sharded layer as threads of ONE process (the form the full-size GPU
    checks use: several processes on one GPU are impractical there) -- same SPMD code, collectives as rendezvous +
    copies, forward / backward driven by hand (tests/sharding_cpu.sharded_magnetic_step)."""
    from pytorch_geometric_signed_directed_amd.parallel import ShardedMagNetConv
    n, f = 60, 16
    g, ei, w = _graph(n, 321, True)
    xr, xi = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
    gr, gi = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
    torch.manual_seed(3)
    weight = torch.empty(k + 1, f, f).uniform_(-0.3, 0.3)
    bias = torch.empty(f).uniform_(-0.5, 0.5)

    def body(rank, exchange):
        layer = ShardedMagNetConv(f, f, k, 0.25, n, ei, w, layout=layout, phases=phases, return_chunks=chunks,
                                  exchange=exchange, kernels=C.KERNELS, operator_rows=C.oracle_operator_rows(ei, w, n, 0.25))
        with torch.no_grad():
            layer.weight.copy_(weight)
            layer.bias.copy_(bias)
        plan = layer.plan
        outs = C.sharded_magnetic_step(layer, plan.shard_rows(xr), plan.shard_rows(xi), plan.shard_rows(gr), plan.shard_rows(gi))
        return plan, outs

    res = C.run_ranks_as_threads(world, body)
    c, d = xr.clone().requires_grad_(), xi.clone().requires_grad_()
    wt, bs = weight.clone().requires_grad_(), bias.clone().requires_grad_()
    op = R.magnet_operator(ei, w, n, 0.25, "sym", 2.0)
    w_r, w_i = R.magnet_conv(c, d, op, wt, bs, duplicate=False)
    ((w_r * gr).sum() + (w_i * gi).sum()).backward()
    want = [w_r.detach(), w_i.detach(), c.grad, d.grad]
    for plan, outs in res:
        for got, ref in zip(outs[:4], want):
            assert float((got[:plan.n_local] - ref[plan.lo:plan.hi]).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
            assert float(got[plan.n_local:].abs().sum()) == 0
        assert float((outs[4] - wt.grad).abs().max()) <= 2e-6 * max(1.0, float(wt.grad.abs().max()))
        assert float((outs[5] - bs.grad).abs().max()) <= 2e-6 * max(1.0, float(bs.grad.abs().max()))
    with pytest.raises(ZeroDivisionError):                  # a failing rank releases the others and is re-raised
        C.run_ranks_as_threads(3, lambda rank, ex: (1 // (rank - 1), ex.all_reduce(torch.ones(1)))[1])


def _digcn_worker(rank, world, port, n, f, phases, block, ret, grid_cols=1, chunks=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // world)))   # the oracle runs in every rank: no oversubscription
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pytorch_geometric_signed_directed_amd.parallel import (ShardedDiGCNConv, ShardedDiGCNInceptionBlock,
                                                                    all_gather_rows)
        C.patch_device_builders()
        g, ei, w = _graph(n, 5, True)
        w = w / 8
        _, ei2, w2 = _graph(n, 6, False)
        w2 = w2 / 8
        x, go = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
        torch.manual_seed(13)
        if block:
            layer = ShardedDiGCNInceptionBlock(f, f, n, ei, w, ei2, w2, phases=phases, kernels=C.KERNELS, grid_cols=grid_cols,
                                               return_chunks=chunks)
        else:
            layer = ShardedDiGCNConv(f, f, n, ei, w, phases=phases, kernels=C.KERNELS, grid_cols=grid_cols, return_chunks=chunks)
        with torch.no_grad():
            for prm in layer.parameters():
                prm.uniform_(-0.5, 0.5)
                dist.broadcast(prm.data, 0)
        plan = layer.plan
        a = layer.shard_rows(x).requires_grad_()
        outs = layer(a)
        outs = outs if block else (outs,)
        loss = sum(((k + 1.0) * o * layer.shard_rows(go)).sum() for k, o in enumerate(outs))
        loss.backward()
        got = [plan.unshard_rows(all_gather_rows(t.detach())) for t in outs + (a.grad,)]
        xo = x.clone().requires_grad_()
        sd = {k: v.detach().clone().requires_grad_() for k, v in layer.named_parameters()}
        if block:
            want = (xo @ sd["ln.weight"].t() + sd["ln.bias"],
                    R.digcn_conv(xo, ei, w, sd["conv1.weight"], sd["conv1.bias"]),
                    R.digcn_conv(xo, ei2, w2, sd["conv2.weight"], sd["conv2.bias"]))
        else:
            want = (R.digcn_conv(xo, ei, w, sd["weight"], sd["bias"]),)
        sum(((k + 1.0) * o * go).sum() for k, o in enumerate(want)).backward()
        worst = 0.0
        pairs = list(zip(got, [t.detach() for t in want] + [xo.grad]))
        pairs += [(prm.grad, sd[k].grad) for k, prm in layer.named_parameters()]
        for x1, y1 in pairs:
            worst = max(worst, float((x1 - y1).abs().max()) / max(1.0, float(y1.abs().max())))
        ret[rank] = worst
    finally:
        dist.destroy_process_group()


def _grad_sync_worker(rank, world, port, ret):
    """Ranks whose autograd graphs DIFFER (rank-dependent extra nodes delay one branch), two backward passes without
    zero_grad: the parameter gradients must still be the un-sharded ones, times two (parallel._GradSync exchanges in
    parameter order at the end of each backward, whatever order autograd produced the gradients in)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pytorch_geometric_signed_directed_amd.parallel import ShardedDiGCNInceptionBlock
        C.patch_device_builders()
        n, f = 40, 8
        g, ei, w = _graph(n, 9, True)
        _, ei2, w2 = _graph(n, 10, False)
        w, w2 = w / 8, w2 / 8
        x, go = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
        torch.manual_seed(3)
        layer = ShardedDiGCNInceptionBlock(f, f, n, ei, w, ei2, w2, kernels=C.KERNELS)
        with torch.no_grad():
            for prm in layer.parameters():
                prm.uniform_(-0.5, 0.5)
                dist.broadcast(prm.data, 0)
        gl = layer.shard_rows(go)

        class _Boom(torch.autograd.Function):              # identity whose backward raises: an aborted backward pass
            @staticmethod
            def forward(ctx, t):
                return t.clone()

            @staticmethod
            def backward(ctx, g_):
                raise ValueError("boom")

        # the input's gradient is the LAST node of the pass: every parameter hook has fired when it raises, and the
        # engine drops the queued end-of-pass callback -- the next passes must exchange as if nothing had happened
        x0, x1, x2 = layer(_Boom.apply(layer.shard_rows(x).requires_grad_()))
        with pytest.raises(ValueError, match="boom"):
            ((x0 * gl).sum() + (x1 * gl).sum() + (x2 * gl).sum()).backward()
        layer.zero_grad(set_to_none=(rank == 0))            # the aborted pass left rank-local sums behind
        for _ in range(2):
            x0, x1, x2 = layer(layer.shard_rows(x))
            if rank == 0:                                   # extra nodes on rank 0 only: conv1's branch becomes ready later
                for _k in range(5):
                    x1 = x1 * 1.0 + 0.0
            else:
                x2 = (x2 + 0.0) * 1.0
            ((x0 * gl).sum() + 2.0 * (x1 * gl).sum() + 3.0 * (x2 * gl).sum()).backward()
        xo = x.clone()
        sd = {k: v.detach().clone().requires_grad_() for k, v in layer.named_parameters()}
        want = (xo @ sd["ln.weight"].t() + sd["ln.bias"], R.digcn_conv(xo, ei, w, sd["conv1.weight"], sd["conv1.bias"]),
                R.digcn_conv(xo, ei2, w2, sd["conv2.weight"], sd["conv2.bias"]))
        ((want[0] * go).sum() + 2.0 * (want[1] * go).sum() + 3.0 * (want[2] * go).sum()).backward()
        ret[rank] = max(float((prm.grad - 2.0 * sd[k].grad).abs().max()) / max(1.0, float(sd[k].grad.abs().max()))
                        for k, prm in layer.named_parameters())
        # torch.autograd.grad would return this rank's share only: refused (after the same collectives on every rank)
        x0, x1, x2 = layer(layer.shard_rows(x))
        with pytest.raises(RuntimeError, match="use .backward"):
            torch.autograd.grad((x0 * gl).sum() + (x1 * gl).sum() + (x2 * gl).sum(), list(layer.parameters()))
    finally:
        dist.destroy_process_group()


def test_parameter_gradients_do_not_depend_on_the_order_autograd_produces_them_in():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_grad_sync_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert len(ret) == 2 and max(ret.values()) <= 4e-6, dict(ret)


def _unused_parameter_worker(rank, world, port, ret):
    """Ranks that use DIFFERENT parameters in one pass, and parameters no rank uses: a parameter some rank used ends the pass
    with the same summed `.grad` on every rank; one that NO rank used keeps `.grad` = None, as on a single device (an optimiser
    skips it -- an explicit zero would still be decayed / given momentum)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pytorch_geometric_signed_directed_amd.parallel import ShardedDiGCNInceptionBlock
        C.patch_device_builders()
        n, f = 40, 8
        g, ei, w = _graph(n, 9, True)
        _, ei2, w2 = _graph(n, 10, False)
        w, w2 = w / 8, w2 / 8
        x, go = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
        torch.manual_seed(3)
        layer = ShardedDiGCNInceptionBlock(f, f, n, ei, w, ei2, w2, kernels=C.KERNELS)
        with torch.no_grad():
            for prm in layer.parameters():
                prm.uniform_(-0.5, 0.5)
                dist.broadcast(prm.data, 0)
        gl = layer.shard_rows(go)
        # pass A: no rank's loss sees the Linear branch -> its parameters keep `.grad` = None on every rank
        layer.zero_grad(set_to_none=True)
        x0, x1, x2 = layer(layer.shard_rows(x))
        (x1 * gl).sum().backward()
        none_ok = layer.ln.weight.grad is None and layer.ln.bias.grad is None and layer.conv1.weight.grad is not None
        # pass B: every rank's loss sees the conv1 branch (its backward is a collective: SPMD), only rank 0's the Linear branch
        # (local rows, no exchange): rank 1 has no share of ln.* but ends the pass with rank 0's
        layer.zero_grad(set_to_none=True)
        x0, x1, x2 = layer(layer.shard_rows(x))
        (((x1 + x0) if rank == 0 else x1) * gl).sum().backward()
        sd = {k: v.detach().clone().requires_grad_() for k, v in layer.named_parameters()}
        lo, hi = layer.plan.bounds[0], layer.plan.bounds[1]
        want_lin = x @ sd["ln.weight"].t() + sd["ln.bias"]
        want_c1 = R.digcn_conv(x, ei, w, sd["conv1.weight"], sd["conv1.bias"])
        # the un-sharded equivalent: conv1's branch on all rows, the Linear on rank 0's rows
        ((want_c1 * go).sum() + (want_lin * go)[lo:hi].sum()).backward()
        worst = 0.0
        for k, prm in layer.named_parameters():
            if not k.startswith("conv2"):          # (conv2 rides in conv1's product node: its share is an explicit zero)
                worst = max(worst, float((prm.grad - sd[k].grad).abs().max()) / max(1.0, float(sd[k].grad.abs().max())))
        ret[rank] = (worst, none_ok)
    finally:
        dist.destroy_process_group()


def test_parameters_no_rank_used_keep_no_gradient_and_partly_used_ones_agree():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_unused_parameter_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert len(ret) == 2, dict(ret)
    for worst, none_ok in ret.values():
        assert worst <= 4e-6 and none_ok, dict(ret)


@pytest.mark.parametrize("world,n,f,phases,block", [(2, 50, 8, 1, False), (3, 61, 4, 2, False), (2, 50, 8, 1, True),
                                                    (4, 90, 8, 2, True)])
def test_sharded_digcn_and_inception_block_equal_unsharded_oracle(world, n, f, phases, block):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_digcn_worker, args=(world, _free_port(), n, f, phases, block, ret), nprocs=world, join=True)
    assert len(ret) == world and max(ret.values()) <= 2e-6, dict(ret)


def _signed_worker(rank, world, port, kind, ret):
    """ShardedSGCNConv (first / deep aggregation) and ShardedSIMPA (undirected / directed, hop 2) against the un-sharded
    oracle layer: outputs, input gradients and the all-reduced parameter gradients (SGCNConv.py:94-126, SIMPA.py:52-144)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // world)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pytorch_geometric_signed_directed_amd.parallel import ShardedSGCNConv, ShardedSIMPA, all_gather_rows
        C.patch_device_builders()
        n, f = 57, 8
        g, pos, w_p = _graph(n, 21, True)
        _, neg, w_n = _graph(n, 22, False)
        neg = neg[:, :200]
        w_n = w_n[:200]
        torch.manual_seed(17)
        if kind.startswith("sgcn"):
            first = kind == "sgcn_first"
            layer = ShardedSGCNConv(f, f // 2, first, n, pos, neg, kernels=C.KERNELS)
            xs = [torch.randn(n, f if first else 2 * f, generator=g)]
        else:
            directed = kind == "simpa_directed"
            layer = ShardedSIMPA(2, 0.5, n, pos, w_p, neg, w_n, directed, kernels=C.KERNELS,
                                 normalise=lambda ei, fill, w, nn_: R.conv_norm_rw(ei, w, nn_, fill))
            xs = [torch.randn(n, f, generator=g) for _ in range(4 if directed else 2)]
        with torch.no_grad():
            for prm in layer.parameters():
                prm.uniform_(0.5, 1.5)
                dist.broadcast(prm.data, 0)
        plan = layer.plan
        local = [layer.shard_rows(x).requires_grad_() for x in xs]
        out = layer(*local)
        go = torch.randn(n, out.size(1), generator=g)
        (out * layer.shard_rows(go)).sum().backward()
        got = [plan.unshard_rows(all_gather_rows(t.detach())) for t in [out] + [a.grad for a in local]]
        ref_in = [x.clone().requires_grad_() for x in xs]
        sd = {k: v.detach().clone().requires_grad_() for k, v in layer.named_parameters()}
        if kind.startswith("sgcn"):
            want = R.sgcn_conv(ref_in[0], pos, neg, (sd["lin_b.weight"], sd["lin_b.bias"]), (sd["lin_u.weight"], sd["lin_u.bias"]),
                               kind == "sgcn_first", f)
        else:
            want = R.simpa(pos, w_p, neg, w_n, ref_in[0], ref_in[1], sd, 2, 0.5, kind == "simpa_directed", *ref_in[2:])
        (want * go).sum().backward()
        pairs = list(zip(got, [want.detach()] + [x.grad for x in ref_in]))
        pairs += [(prm.grad, sd[k].grad) for k, prm in layer.named_parameters()]
        ret[rank] = max(float((a - b).abs().max()) / max(1.0, float(b.abs().max())) for a, b in pairs)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "sgcn_first"), (3, "sgcn_deep"), (2, "simpa_undirected"), (3, "simpa_directed")])
def test_sharded_signed_layers_equal_the_unsharded_oracle(world, kind):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_signed_worker, args=(world, _free_port(), kind, ret), nprocs=world, join=True)
    assert len(ret) == world and max(ret.values()) <= 4e-6, dict(ret)


@pytest.mark.parametrize("world,n,f,grid_cols,chunks,block", [(4, 90, 8, 2, 1, False), (4, 90, 16, 4, 2, True), (8, 130, 8, 2, 2, True)])
def test_sharded_digcn_in_the_grid_layout(world, n, f, grid_cols, chunks, block):
    """One-operand operators in the p_r x p_c process grid (round 4): column-slice all-to-all in, row-chunked all-to-all back."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_digcn_worker, args=(world, _free_port(), n, f, 1, block, ret, grid_cols, chunks), nprocs=world, join=True)
    assert len(ret) == world and max(ret.values()) <= 2e-6, dict(ret)


@pytest.mark.parametrize("world,p_c,phases,chunks,f", [(8, 4, (0.4, 0.6), (0.5, 0.36, 0.14), 64), (8, 4, 2, 2, 128),
                                                       (4, 2, (0.3, 0.3, 0.4), 1, 64), (2, 1, 2, 1, 64)])
def test_piece_layouts_address_what_packing_and_merging_produce(world, p_c, phases, chunks, f):
    """Round 5: the dense kernels write the send buffers / read the receive buffer of the exchanges THEMSELVES, through
    pygsd_piece_layout.  The layouts the engine hands them (`send_layout`, `return_layout`) are held here to the tensor-op
    packing and merging they replace, through the layout formula restated in `_cabi.PieceLayout.offsets` (the formula the HIP
    kernels implement; tests/test_gpu_sharded.py holds the kernels to the same)."""
    grid = p_c > 1
    align = PropagateEngine.alignment(world, p_c, phases, chunks)
    plan = ShardPlan(997, world, world - 1, align=align)
    eng = PropagateEngine(plan, type("Ex", (), {"world_size": world, "rank": world - 1})(), p_c, phases, chunks, C.KERNELS)
    g = torch.Generator().manual_seed(7)
    xs = [torch.randn(plan.n_pad, f, generator=g) for _ in range(2)]
    layout, bufs = eng.send_layout(2, f, xs[0])
    fw = f // p_c
    assert layout.slot_floats == fw and layout.replicas == (eng.p_r if grid else 1)
    esz = 4
    for rep in range(layout.replicas):
        off = layout.offsets(plan.n_pad, f, rep)
        for c in range(eng.phases):
            packed = eng._pack(xs, c).reshape(-1)
            assert packed.data_ptr() == bufs[c].data_ptr()
            rows = slice(eng.phase_bounds[c], eng.phase_bounds[c + 1])
            local = off[rows] - layout.base[c]
            assert int(local.min()) >= 0 and int(local.max()) + fw < packed.numel() + fw
            for grp in range(2):
                assert torch.equal(packed[local + grp * fw], xs[grp][rows])
        assert layout.base[c] == (bufs[c].data_ptr() - bufs[0].data_ptr()) // esz
    if grid:
        recv = torch.randn(eng.block_rows, 2 * fw, generator=g)
        merged = eng._merge(recv, 2)
        rl = eng.return_layout(2, fw)
        off = rl.offsets(plan.n_pad, f)
        for grp in range(2):
            assert torch.equal(recv.reshape(-1)[off + grp * fw], merged[grp])
        st = rl.struct()
        assert st.n_chunks == eng.return_chunks and st.blk_rows == eng.n_blk and list(st.lo)[:eng.return_chunks + 1] == eng.chunk_bounds


def _cached_input_worker(rank, world, port, ret):
    """cache_input_exchange=True: the second forward over the same (unmodified) feature tensors must not exchange them again and
    must give the same outputs and gradients; an in-place edit of the features bumps their version and is seen."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pytorch_geometric_signed_directed_amd.parallel import ShardedMagNetConv, all_gather_rows
        n, f, k = 61, 16, 2
        g, ei, w = _graph(n, 5, True)
        xr, xi = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
        gr, gi = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
        torch.manual_seed(3)
        layer = ShardedMagNetConv(f, f, k, 0.25, n, ei, w, layout="grid", phases=(0.4, 0.6), return_chunks=2, kernels=C.KERNELS,
                                  operator_rows=C.oracle_operator_rows(ei, w, n, 0.25), cache_input_exchange=True)
        calls = {"n": 0}
        inner = layer.engine.ex.all_to_all

        def counting(out, inp):
            calls["n"] += 1
            return inner(out, inp)
        layer.engine.ex.all_to_all = counting
        plan = layer.plan
        a, b = layer.shard_rows(xr).requires_grad_(), layer.shard_rows(xi).requires_grad_()

        def step():
            layer.zero_grad(set_to_none=True)
            a.grad = b.grad = None
            o_r, o_i = layer(a, b)
            ((o_r * layer.shard_rows(gr)).sum() + (o_i * layer.shard_rows(gi)).sum()).backward()
            return [plan.unshard_rows(all_gather_rows(t.detach())) for t in (o_r, o_i, a.grad, b.grad)]

        def oracle(x_r, x_i):
            c, d = x_r.clone().requires_grad_(), x_i.clone().requires_grad_()
            op = R.magnet_operator(ei, w, n, 0.25, "sym", 2.0)
            w_r, w_i = R.magnet_conv(c, d, op, layer.weight.detach(), layer.bias.detach(), duplicate=False)
            ((w_r * gr).sum() + (w_i * gi).sum()).backward()
            return [w_r.detach(), w_i.detach(), c.grad, d.grad]

        def err(got, want):
            return max(float((x - y).abs().max()) / max(1.0, float(y.abs().max())) for x, y in zip(got, want))

        first = step()
        per_step = calls["n"]                      # K = 2: 2 forward + 2 backward propagates x (2 phases in + 2 chunks back)
        second = step()
        saved = 2 * per_step - calls["n"]
        assert saved == layer.engine.phases, (per_step, calls["n"])     # exactly ONE propagate's inbound phases were not repeated
        worst = max(err(first, oracle(xr, xi)), err(second, oracle(xr, xi)))
        with torch.no_grad():
            a.mul_(1.5)                            # in-place: the version moves, the memo must miss
        before = calls["n"]
        third = step()
        assert calls["n"] - before == per_step
        worst = max(worst, err(third, oracle(1.5 * xr, xi)))
        ret[rank] = worst
    finally:
        dist.destroy_process_group()


def test_cached_input_exchange_is_exact_and_sees_in_place_edits():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_cached_input_worker, args=(4, _free_port(), ret), nprocs=4, join=True)
    assert len(ret) == 4 and max(ret.values()) <= 2e-6, dict(ret)
