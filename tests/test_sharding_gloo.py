"""CPU, world_size 2 and 3, gloo: the node-range sharding plan and the all-gather exchange of
pytorch_geometric_signed_directed_amd.parallel.  Each rank keeps only the operator rows it
produces, gathers the packed (real | imag) feature blocks, evaluates its local rows with the ORACLE
(the HIP kernels cannot run here) and must reproduce exactly the rows of the un-sharded oracle
result -- forward and backward (dX = by-source rows x gathered upstream gradient)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_layers as R
from pytorch_geometric_signed_directed_amd.parallel import ShardPlan, all_gather_rows, pack_pair


def test_plan_partitions_every_node_once():
    for n, w in [(10, 2), (11, 3), (7, 8), (1000, 8), (1, 2)]:
        seen = []
        for r in range(w):
            p = ShardPlan(n, w, r)
            assert p.n_total >= n and p.n_total - n < w
            seen += list(range(p.lo, p.hi))
            x = torch.arange(n, dtype=torch.float32).unsqueeze(1)
            s = p.shard_rows(x)
            assert s.shape == (p.n_pad, 1)
            assert s[:p.n_local, 0].tolist() == list(range(p.lo, p.hi)) and float(s[p.n_local:].abs().sum()) == 0
        assert seen == list(range(n))


def test_local_entries_rebase():
    ei = torch.tensor([[0, 5, 9, 3, 7], [9, 0, 4, 3, 8]])
    p = ShardPlan(10, 2, 1)                      # owns [5, 10)
    keep, sub = p.local_entries(ei, by=1)
    assert keep.tolist() == [0, 4] and sub.tolist() == [[0, 7], [4, 3]]
    keep, sub = p.local_entries(ei, by=0)
    assert keep.tolist() == [1, 2, 4] and sub.tolist() == [[0, 4, 2], [0, 4, 8]]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        g = torch.Generator().manual_seed(123)
        e, f = 12 * n, 5
        ei = torch.randint(0, n, (2, e), generator=g)
        w = torch.rand(e, generator=g) + 0.5
        xr, xi = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
        gr, gi = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
        plan = ShardPlan(n, world, rank)
        # un-sharded oracle (reference op sequence) incl. gradients
        op = R.magnet_operator(ei, w, n, 0.25, "sym", 2.0)
        a, b = xr.clone().requires_grad_(), xi.clone().requires_grad_()
        t_r = R.propagate(a, op[0], op[2], n)
        t_i = R.propagate(b, op[1], op[3], n)
        ((t_r * gr).sum() + (t_i * gi).sum()).backward()
        # ---- sharded forward: gather packed blocks, evaluate owned target rows only
        full = all_gather_rows(pack_pair(plan.shard_rows(xr), plan.shard_rows(xi)))
        assert full.shape == (plan.n_total, 2 * f)
        assert torch.equal(plan.unshard_rows(full), torch.cat([xr, xi], dim=1))
        outs = []
        for op_index, op_val, cols in ((op[0], op[2], slice(0, f)), (op[1], op[3], slice(f, 2 * f))):
            keep, sub = plan.local_entries(op_index, by=1)
            outs.append(R.propagate(full[:, cols], sub, op_val[keep], plan.n_pad))
        ok = torch.equal(outs[0][:plan.n_local], t_r.detach()[plan.lo:plan.hi]) and \
            torch.equal(outs[1][:plan.n_local], t_i.detach()[plan.lo:plan.hi])
        ok = ok and float(outs[0][plan.n_local:].abs().sum()) == 0.0
        # ---- sharded backward: gather the upstream gradient, evaluate owned SOURCE rows
        gfull = all_gather_rows(pack_pair(plan.shard_rows(gr), plan.shard_rows(gi)))
        grads = []
        for op_index, op_val, cols in ((op[0], op[2], slice(0, f)), (op[1], op[3], slice(f, 2 * f))):
            keep, sub = plan.local_entries(op_index, by=0)
            # dX[src] += w * dT[tgt]: gather at row 1 (targets, global), scatter at row 0 (sources, local)
            grads.append(R.propagate(gfull[:, cols], sub, op_val[keep], plan.n_pad, flow="target_to_source"))
        ok = ok and torch.allclose(grads[0][:plan.n_local], a.grad[plan.lo:plan.hi], rtol=0, atol=1e-6)
        ok = ok and torch.allclose(grads[1][:plan.n_local], b.grad[plan.lo:plan.hi], rtol=0, atol=1e-6)
        # ---- parameter-gradient style all-reduce: per-shard partial sums add up to the global sum
        part = outs[0][:plan.n_local].sum(0)
        dist.all_reduce(part)
        ok = ok and torch.allclose(part, t_r.detach().sum(0), rtol=1e-5, atol=1e-5)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 40), (3, 41)])
def test_sharded_rows_equal_unsharded_oracle(world, n):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


# ------------------------------------------------------------------------------------------------
# grid layout (GridPlan): column-slice exchange, (row block i) x (column slice j) product, exchange back
# ------------------------------------------------------------------------------------------------
def test_grid_plan_geometry():
    from pytorch_geometric_signed_directed_amd.parallel import GridPlan
    assert [GridPlan.choose_cols(w, 64) for w in (1, 2, 4, 6, 8)] == [1, 1, 4, 2, 4]
    assert GridPlan.choose_cols(8, 12) == 1 and GridPlan.choose_cols(8, 24) == 2      # 16-byte aligned slices only
    p = GridPlan(1000, 8, 5, 64)
    assert (p.p_r, p.p_c, p.i, p.j, p.fc) == (2, 4, 1, 1, 16)
    assert p.block_rows == 4 * p.n_pad and p.block_lo == 4 * p.n_pad and list(p.row_group()) == [4, 5, 6, 7]
    assert p.group_splits() == [0, 0, 0, 0, 1, 1, 1, 1]
    a = torch.arange(p.n_pad * 64, dtype=torch.float32).view(p.n_pad, 64)
    b = -a
    chunks = p.slice_chunks(a, b)
    assert chunks.shape == (8, p.n_pad, 32)
    for d in range(8):
        j = d % 4
        assert torch.equal(chunks[d][:, :16], a[:, 16 * j:16 * j + 16]) and torch.equal(chunks[d][:, 16:], b[:, 16 * j:16 * j + 16])
    recv = torch.stack([chunks[j] for j in range(4)])          # what a row group hands back for these rows
    ra, rb = p.merge_slices(recv)
    assert torch.equal(ra, a) and torch.equal(rb, b)
    with pytest.raises(ValueError):
        GridPlan(10, 6, 0, 64, 4)


def _grid_worker(rank, world, port, n, f, ret):
    from pytorch_geometric_signed_directed_amd.parallel import GridPlan, collect_slices, return_rows
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(321)
        e = 10 * n
        ei = torch.randint(0, n, (2, e), generator=g)
        w = torch.rand(e, generator=g) + 0.5
        xr, xi = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
        plan = GridPlan(n, world, rank, f, 2 if world == 2 else None)      # two ranks: force the 1 x 2 grid
        fc = plan.fc
        op = R.magnet_operator(ei, w, n, 0.25, "sym", 2.0)
        t_r = R.propagate(xr, op[0], op[2], n)                           # un-sharded oracle
        t_i = R.propagate(xi, op[1], op[3], n)
        full = collect_slices(plan, plan.shard_rows(xr), plan.shard_rows(xi))
        cols = slice(plan.j * fc, (plan.j + 1) * fc)
        ok = full.shape == (plan.n_total, 2 * fc)
        ok = ok and torch.equal(full[:n, :fc], xr[:, cols]) and torch.equal(full[:n, fc:], xi[:, cols])
        ok = ok and float(full[n:].abs().sum()) == 0.0
        # product of operator row block i with column slice j, by the oracle
        ys = []
        for op_index, op_val, part in ((op[0], op[2], full[:, :fc]), (op[1], op[3], full[:, fc:])):
            tgt = op_index[1]
            keep = ((tgt >= plan.block_lo) & (tgt < plan.block_lo + plan.block_rows)).nonzero(as_tuple=True)[0]
            sub = op_index[:, keep].clone()
            sub[1] -= plan.block_lo
            ys.append(R.propagate(part, sub, op_val[keep], plan.block_rows))
        ra, rb = return_rows(plan, ys[0], ys[1])
        ok = ok and ra.shape == (plan.n_pad, f)
        ok = ok and torch.equal(ra[:plan.n_local], t_r[plan.lo:plan.hi]) and torch.equal(rb[:plan.n_local], t_i[plan.lo:plan.hi])
        ok = ok and float(ra[plan.n_local:].abs().sum()) == 0.0
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,f", [(2, 40, 8), (4, 41, 16), (8, 50, 16), (3, 30, 8)])
def test_grid_exchanges_reproduce_unsharded_oracle_rows(world, n, f):
    """2 = 1x2, 4 = 1x4, 8 = 2x4 grids, and 3 ranks (no column split possible: 3x1)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_grid_worker, args=(world, _free_port(), n, f, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}
