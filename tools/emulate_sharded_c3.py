"""Rehearse ONE rank of BASELINE config C3 in its sharded form -- ShardedSGCNConv (first aggregation, 64 -> 32) and ShardedSIMPA
(hop 2, SSSNET's aggregation) on SSBM 500k nodes / 10M +- entries, 8 ranks, row layout (one all-gather per propagate) -- on a
single MI355X, exchanges played by parallel.EmulatedExchange (tools/emulate_sharded.py says what is real and what is played).
Next to each: the same layer un-sharded on this GPU, timed in the same process (the denominator of the projected speed-up)."""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(step, steps):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); step(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--link-gbps", type=float, default=61.0)
    ap.add_argument("--nodes", type=int, default=500000)
    ap.add_argument("--entries", type=int, default=10000000)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    from pytorch_geometric_signed_directed_amd import graphs
    from pytorch_geometric_signed_directed_amd.nn import SGCNConv, SIMPA
    from pytorch_geometric_signed_directed_amd.parallel import EmulatedExchange, ShardedSGCNConv, ShardedSIMPA
    dev = torch.device("cuda:0")
    n, h = args.nodes, args.hidden
    p = (args.entries / 2) / (n * (n - 1) / 2)
    ei_np, sign, _ = graphs.ssbm(n, 5, p, 0.1, 2.0, seed=2)
    ei, sign = torch.from_numpy(ei_np).to(dev), torch.from_numpy(sign).to(dev)
    pos, neg = ei[:, sign > 0].contiguous(), ei[:, sign < 0].contiguous()
    wp, wn = torch.ones(pos.size(1), device=dev), torch.ones(neg.size(1), device=dev)
    g = torch.Generator().manual_seed(0)
    x, xp, xn = (torch.randn(n, h, generator=g).to(dev) for _ in range(3))
    out = {"world": args.world, "nodes": n, "pos_entries": int(pos.size(1)), "neg_entries": int(neg.size(1)), "hidden": h,
           "link_gbps": args.link_gbps, "runs": {}}

    def sharded(make, inputs, forward):
        ex = EmulatedExchange(args.world, 0, args.link_gbps)
        torch.manual_seed(0)
        layer = make(ex)
        loc = [layer.shard_rows(t).requires_grad_() for t in inputs]

        def step():
            layer.zero_grad(set_to_none=True)
            for t in loc:
                t.grad = None
            forward(layer, loc).sum().backward()
        ms = timed(step, args.steps)
        layer.engine.profile(True)
        ex.wire_us = 0.0
        for _ in range(args.steps):
            step()
        summary = layer.engine.timing_summary()
        props = summary["propagates"] / args.steps
        rec = {"step_ms_median": ms, "propagates_per_step": props, "per_propagate": summary,
               "wire_ms_per_propagate": ex.wire_us / 1e3 / max(summary["propagates"], 1), "n_pad": layer.plan.n_pad}
        del layer
        torch.cuda.empty_cache()
        return rec

    def single(layer, inputs, forward):
        loc = [t.clone().requires_grad_() for t in inputs]

        def step():
            layer.zero_grad(set_to_none=True)
            for t in loc:
                t.grad = None
            forward(layer, loc).sum().backward()
        return timed(step, args.steps)

    torch.manual_seed(0)
    one = single(SGCNConv(h, h // 2, first_aggr=True).to(dev), [x], lambda ly, t: ly(t[0], pos, neg))
    rec = sharded(lambda ex: ShardedSGCNConv(h, h // 2, True, n, pos, neg, device=dev, exchange=ex), [x], lambda ly, t: ly(t[0]))
    rec.update(single_gpu_ms=one, projected_speedup=one / rec["step_ms_median"])
    out["runs"]["sgcnconv_first"] = rec
    print("sgcnconv_first", json.dumps(rec), flush=True)
    one = single(SIMPA(2, 0.5).to(dev), [xp, xn], lambda ly, t: ly(pos, wp, neg, wn, t[0], t[1]))
    rec = sharded(lambda ex: ShardedSIMPA(2, 0.5, n, pos, wp, neg, wn, device=dev, exchange=ex), [xp, xn], lambda ly, t: ly(t[0], t[1]))
    rec.update(single_gpu_ms=one, projected_speedup=one / rec["step_ms_median"])
    out["runs"]["simpa_hop2"] = rec
    print("simpa_hop2", json.dumps(rec), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "emulated_sharded_c3.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
