"""Rehearse ONE rank of BASELINE config C5 -- DiGCN_InceptionBlock, 2M nodes / 52M entries per operator, bf16 (and
fp32), 8 ranks, row layout with ONE all-gather per propagate for both convolutions -- on a single MI355X, exchanges
played by parallel.EmulatedExchange (see tools/emulate_sharded.py for what is real and what is played)."""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--link-gbps", type=float, default=61.0)
    ap.add_argument("--nodes", type=int, default=2000000)
    ap.add_argument("--edges", type=int, default=25000000)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--grid-cols", type=int, default=1, help="p_c of the p_r x p_c process grid (1 = row layout)")
    ap.add_argument("--return-chunks", type=int, default=1)
    ap.add_argument("--phases", type=int, default=1)
    ap.add_argument("--single-gpu-ms", default="9.80,5.30", help="fp32,bf16 block step on one GPU (profiles/r4j_configs.json)")
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    from pytorch_geometric_signed_directed_amd import graphs
    from pytorch_geometric_signed_directed_amd.parallel import EmulatedExchange, ShardedDiGCNInceptionBlock
    dev = torch.device("cuda:0")
    n, h = args.nodes, args.hidden
    src, dst = torch.from_numpy(graphs.dsbm_for_edges(n, args.edges, seed=3)[0]).to(dev)
    loops = torch.arange(n, device=dev)
    ops = []
    for k in range(2):           # two symmetric, positively weighted, sym-normalised operators (as tools/bench_configs.py)
        g = torch.Generator(device="cuda").manual_seed(10 + k)
        d2 = dst if k == 0 else dst[torch.randperm(dst.numel(), device=dev, generator=g)]
        wv = torch.rand(src.numel(), device=dev, generator=g)
        ei = torch.stack([torch.cat([src, d2, loops]), torch.cat([d2, src, loops])])
        w = torch.cat([wv, wv, torch.ones(n, device=dev)])
        deg = torch.zeros(n, device=dev).index_add_(0, ei[0], w)
        ops.append((ei, deg[ei[0]].rsqrt() * w * deg[ei[1]].rsqrt()))
    out = {"world": args.world, "nodes": n, "entries_per_operator": int(ops[0][0].size(1)), "hidden": h,
           "link_gbps": args.link_gbps, "grid_cols": args.grid_cols, "return_chunks": args.return_chunks, "phases": args.phases,
           "single_gpu_ms": dict(zip(("float32", "bfloat16"), (float(v) for v in args.single_gpu_ms.split(",")))), "runs": {}}
    for dtype in (torch.float32, torch.bfloat16):
        ex = EmulatedExchange(args.world, 0, args.link_gbps)
        torch.manual_seed(0)
        block = ShardedDiGCNInceptionBlock(h, h, n, ops[0][0], ops[0][1], ops[1][0], ops[1][1], device=dev, exchange=ex,
                                           grid_cols=args.grid_cols, return_chunks=args.return_chunks, phases=args.phases)
        block.to(dtype)
        x = block.shard_rows(torch.randn(n, h, device=dev)).to(dtype).requires_grad_()

        def step():
            block.zero_grad(set_to_none=True)
            x.grad = None
            x0, x1, x2 = block(x)
            (x0 + x1 + x2).float().sum().backward()

        for _ in range(3):
            step()
        ts = []
        for _ in range(args.steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); step(); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
        block.engine.profile(True)
        ex.wire_us = 0.0
        for _ in range(args.steps):
            step()
        name = str(dtype).split(".")[-1]
        rec = {"step_ms_median": statistics.median(ts), "per_propagate": block.engine.timing_summary(),
               "wire_ms_per_propagate": ex.wire_us / 1e3 / (2 * args.steps)}
        rec["projected_speedup"] = out["single_gpu_ms"][name] / rec["step_ms_median"]
        out["runs"][name] = rec
        print(name, json.dumps(rec), flush=True)
        del block, x
        torch.cuda.empty_cache()
    with open(os.path.join(ROOT, "gpurun_out", f"emulated_sharded_c5{args.tag}.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
