"""GPU: ShardedMagNetConv (HIP compute, node-range ownership; row layout and p_r x p_c grid layout) against
the oracle.  2 .. 8 ranks share the one GPU of the test box, exchanging through gloo (RCCL refuses two ranks
on one device; the 8-GPU RCCL run is the driver's) -- the compute path, packing / slicing, operator
blocks, the Chebyshev adjoint over exchanged blocks and the parameter all-reduce are the production code."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, k, f, layout, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref_layers as R
        from pytorch_geometric_signed_directed_amd.parallel import ShardedMagNetConv, all_gather_rows
        dev = torch.device("cuda:0")
        g = torch.Generator().manual_seed(7)
        e = 15 * n
        ei = torch.randint(0, n, (2, e), generator=g)
        w = torch.rand(e, generator=g) + 0.5
        xr, xi = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
        gr, gi = torch.randn(n, f, generator=g), torch.randn(n, f, generator=g)
        torch.manual_seed(11)
        layer = ShardedMagNetConv(f, f, k, 0.25, n, ei.to(dev), w.to(dev), device=dev, layout=layout,
                                  grid_cols=2 if (layout == "grid" and world == 2) else None)   # force the 1 x 2 grid
        assert layer.layout == layout and (layout == "rows" or layer.plan.p_c > 1)
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
        op = R.magnet_operator(ei, w, n, 0.25, "sym", 2.0)
        w_r, w_i = R.magnet_conv(c, d, op, wt, bs, duplicate=False)
        ((w_r * gr).sum() + (w_i * gi).sum()).backward()
        want = [w_r.detach(), w_i.detach(), c.grad, d.grad]
        worst = 0.0
        for x, y in zip(got, want):
            worst = max(worst, float((x - y).abs().max()) / max(1.0, float(y.abs().max())))
        for x, y in ((layer.weight.grad.cpu(), wt.grad), (layer.bias.grad.cpu(), bs.grad)):
            worst = max(worst, float((x - y).abs().max()) / max(1.0, float(y.abs().max())))
        ret[rank] = worst
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,k,f,layout", [
    (2, 1003, 2, 64, "rows"), (3, 500, 3, 16, "rows"), (2, 300, 2, 6, "rows"),
    (2, 1003, 1, 64, "grid"),          # 1 x 2
    (4, 901, 3, 32, "grid"),           # 1 x 4 at 8-float slices, three Chebyshev orders (z term in the grid layout)
    (8, 1203, 2, 64, "grid"),          # 2 x 4: the 8-GPU configuration
])
def test_sharded_layer_matches_oracle(world, n, k, f, layout):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, k, f, layout, ret), nprocs=world, join=True)
    assert len(ret) == world
    assert max(ret.values()) <= 1e-5, dict(ret)


def _digcn_worker(rank, world, port, n, f, dtype_name, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref_layers as R
        from pytorch_geometric_signed_directed_amd.parallel import ShardedDiGCNConv, all_gather_rows
        dev = torch.device("cuda:0")
        dtype = getattr(torch, dtype_name)
        g = torch.Generator().manual_seed(5)
        e = 12 * n
        ei = torch.randint(0, n, (2, e), generator=g)
        w = torch.rand(e, generator=g) / 8
        x = torch.randn(n, f, generator=g)
        go = torch.randn(n, f, generator=g)
        torch.manual_seed(13)
        layer = ShardedDiGCNConv(f, f, n, ei.to(dev), w.to(dev), device=dev)
        with torch.no_grad():
            layer.bias.uniform_(-0.5, 0.5)
            dist.broadcast(layer.bias.data, 0)
        wt, bs = layer.weight.detach().cpu().clone(), layer.bias.detach().cpu().clone()
        layer.to(dtype)
        a = layer.shard_rows(x.to(dev)).to(dtype).requires_grad_()
        out = layer(a)
        (out.float() * layer.shard_rows(go.to(dev))).sum().backward()
        got = [layer.plan.unshard_rows(all_gather_rows(t.detach().float().contiguous())).cpu() for t in (out, a.grad)]
        # oracle (fp32 on inputs rounded to the storage dtype)
        rnd = (lambda t: t.to(dtype).float())
        xo, wo = rnd(x).requires_grad_(), rnd(wt).requires_grad_()
        want = R.digcn_conv(xo, ei, w, wo, rnd(bs))
        (want * go).sum().backward()
        tol = 1e-5 if dtype is torch.float32 else 3e-2
        worst = 0.0
        for x1, y1 in zip(got, (want.detach(), xo.grad)):
            worst = max(worst, float((x1 - y1).abs().max()) / max(1.0, float(y1.abs().max())))
        worst = max(worst, float((layer.weight.grad.float().cpu() - wo.grad).abs().max()) /
                    max(1.0, float(wo.grad.abs().max())))
        ret[rank] = worst / tol
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,f,dtype_name", [(2, 1001, 64, "float32"), (3, 600, 16, "float32"),
                                                   (2, 1000, 64, "bfloat16")])
def test_sharded_digcn_matches_oracle(world, n, f, dtype_name):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_digcn_worker, args=(world, _free_port(), n, f, dtype_name, ret), nprocs=world, join=True)
    assert len(ret) == world
    assert max(ret.values()) <= 1.0, dict(ret)
