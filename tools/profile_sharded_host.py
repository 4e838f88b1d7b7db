"""Where the HOST spends the time it takes to issue one sharded MagNetConv step (one rank of a P-rank job rehearsed on one GPU,
parallel.EmulatedExchange): cProfile over steps issued back to back without synchronisation.

    python tools/profile_sharded_host.py [--world 8] [--steps 50] [--top 45]
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--nodes", type=int, default=1000000)
    ap.add_argument("--edges", type=int, default=20000000)
    ap.add_argument("--hidden", type=int, default=64)
    args = ap.parse_args()
    from pytorch_geometric_signed_directed_amd import graphs
    from pytorch_geometric_signed_directed_amd.parallel import EmulatedExchange, ShardedMagNetConv
    dev = torch.device("cuda:0")
    ei = torch.from_numpy(graphs.dsbm_for_edges(args.nodes, args.edges, seed=0)[0]).to(dev)
    g = torch.Generator().manual_seed(0)
    x_real = torch.randn(args.nodes, args.hidden, generator=g).to(dev)
    x_imag = torch.randn(args.nodes, args.hidden, generator=g).to(dev)
    ex = EmulatedExchange(args.world, 0)
    torch.manual_seed(0)
    layer = ShardedMagNetConv(args.hidden, args.hidden, 1, 0.25, args.nodes, ei, None, device=dev, exchange=ex)
    xr = layer.shard_rows(x_real).requires_grad_()
    xi = layer.shard_rows(x_imag).requires_grad_()

    def step():
        layer.zero_grad(set_to_none=True)
        xr.grad = xi.grad = None
        o_r, o_i = layer(xr, xi)
        (o_r.sum() + o_i.sum()).backward()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_ms = (time.perf_counter() - t0) / args.steps * 1e3
    torch.cuda.synchronize()
    print(f"host issue {host_ms:.3f} ms / step (layout {layer.layout}, p_r x p_c = {layer.engine.p_r} x {layer.engine.p_c})")
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(args.steps):
        step()
    prof.disable()
    torch.cuda.synchronize()
    out = io.StringIO()
    st = pstats.Stats(prof, stream=out)
    st.sort_stats("cumulative").print_stats(args.top)
    print(out.getvalue())
    out = io.StringIO()
    st = pstats.Stats(prof, stream=out)
    st.sort_stats("tottime").print_stats(30)
    print(out.getvalue())


if __name__ == "__main__":
    main()
