"""Rehearse ONE rank of a P-rank sharded MagNetConv step on a single MI355X (no multi-GPU box is available to
this build; the driver's 8-GPU run is the real measurement).

    python tools/emulate_sharded.py [--world 8] [--rank 0] [--link-gbps 61] [--nodes 1000000] [--edges 20000000]

What is real: the plan (equal-work ranges), this rank's operator rows at their real size (1 / p_r of the
north-star operator, split into column phases), every kernel it would launch (partial dual SpMMs at the sliced
width, the fused dense stage on its n_pad rows, packing / merging), the stream / event pipeline of
parallel.PropagateEngine.  What is played: each exchange is a device copy of the bytes this rank would receive
plus a timed kernel of the wire time of the busiest link (bytes_per_link / link rate + a fixed latency) on a
separate stream (parallel.EmulatedExchange).  Received values are stand-ins, so outputs are not checked here
(tests/test_gpu_sharded.py does that over gloo).

For each pipeline shape it reports the step time, the per-propagate split (product / exposed exchange / pack /
merge), the emulated wire time and the overlap fraction = 1 - exposed / wire.  The projection for P GPUs is this
rank's step time (ranks are symmetric on a balanced SBM): speed-up = single-GPU step / emulated step.
"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_shape(args, edge_index, x_real, x_imag, layout, phases, chunks, link_gbps):
    from pytorch_geometric_signed_directed_amd.parallel import EmulatedExchange, ShardedMagNetConv
    dev = x_real.device
    ex = EmulatedExchange(args.world, args.rank, link_gbps, args.latency_us)
    torch.manual_seed(0)
    layer = ShardedMagNetConv(args.hidden, args.hidden, args.K, 0.25, args.nodes, edge_index, args.edge_weight, device=dev,
                              layout=layout, phases=phases, return_chunks=chunks, exchange=ex, signed=args.signed,
                              cache_input_exchange=args.cache_input_exchange)
    xr = layer.shard_rows(x_real).requires_grad_()
    xi = layer.shard_rows(x_imag).requires_grad_()

    def step():
        layer.zero_grad(set_to_none=True)
        xr.grad = xi.grad = None
        o_r, o_i = layer(xr, xi)
        (o_r.sum() + o_i.sum()).backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    times = []
    for _ in range(args.steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        step()
        b.record()
        b.synchronize()
        times.append(a.elapsed_time(b))
    # how long the HOST takes to issue one step (no synchronisation inside the loop): a step of ~2 ms made of ~60 launches, events
    # and stream switches is paced by the host where this approaches the device time
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_ms = (time.perf_counter() - t0) / args.steps * 1e3
    torch.cuda.synchronize()
    layer.engine.profile(True)
    ex.wire_us = 0.0
    for _ in range(args.steps):
        step()
    summary = layer.engine.timing_summary()
    layer.engine.profile(False)
    wire_ms = ex.wire_us / 1e3 / (2 * args.K * args.steps)         # per propagate
    # the same step replayed from a hipGraph (hipgraph.capture_step: exchanges on their side stream, played wire times and all):
    # what the schedule costs the DEVICE once the host is out of the loop
    graph_ms = graph_err = None
    if not args.no_graph:
        try:
            from pytorch_geometric_signed_directed_amd.hipgraph import capture_step
            replay = capture_step(step, warmup=2)
            for _ in range(3):
                replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(args.steps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                replay()
                b.record()
                b.synchronize()
                ts.append(a.elapsed_time(b))
            graph_ms = statistics.median(ts)
            del replay
        except Exception as exc:  # noqa: BLE001 -- a capture that fails must not cost the eager numbers
            graph_err = repr(exc)[:300]
            torch.cuda.synchronize()
    eng = layer.engine
    rec = {"layout": layout, "p_r": eng.p_r, "p_c": eng.p_c, "phases": phases, "return_chunks": chunks,
           "link_gbps": link_gbps, "step_ms_median": statistics.median(times), "step_ms_min": min(times),
           "host_issue_ms_per_step": host_ms, "step_ms_hipgraph_replay": graph_ms, "hipgraph_error": graph_err,
           "per_propagate": summary, "wire_ms_per_propagate": wire_ms,
           "overlap_fraction": (1.0 - summary["exposed_exchange_ms"] / wire_ms) if wire_ms > 0 else None,
           "phase_rows": eng.phase_rows, "return_chunk_rows": eng.chunk_rows, "merge_on_read": bool(getattr(layer, "merge_on_read", False)),
           "packed_backward": bool(getattr(layer, "packed_backward", False)),
           "cache_input_exchange": bool(args.cache_input_exchange),
           "local_operator_entries": layer.local_nnz, "rows_multiplied": eng.block_rows, "n_pad": layer.plan.n_pad}
    del layer
    torch.cuda.empty_cache()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--link-gbps", type=float, nargs="+", default=[61.0],
                    help="emulated rate of one xGMI link, one direction (76.8 GB/s peak x 0.8)")
    ap.add_argument("--latency-us", type=float, default=10.0)
    ap.add_argument("--nodes", type=int, default=1000000)
    ap.add_argument("--edges", type=int, default=20000000)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--K", type=int, default=1, help="Chebyshev order (2 K propagates per step)")
    ap.add_argument("--signed", action="store_true", help="MSConv on an SDSBM graph (BASELINE config C4 with --hidden 128 --K 2)")
    ap.add_argument("--shapes", nargs="+", default=None, metavar="LAYOUT:C:R",
                    help="pipeline shapes to run, e.g. grid:2:2 rows:1:1 -- C / R a count (equal pieces) or fractions, "
                         "grid:0.4,0.6:0.5,0.36,0.14 (default: the built-in sweep)")
    ap.add_argument("--cache-input-exchange", action="store_true",
                    help="opt-in of the layer: the forward propagate's INBOUND exchange is memoised while x_real / x_imag are the same "
                         "tensors at the same version (input features of a first layer: they do not change between steps)")
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph-replayed variant of every shape")
    ap.add_argument("--single-gpu-ms", type=float, default=None, help="measured 1-GPU step (bench.py) for the ratio")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "emulated_sharded.json"))
    args = ap.parse_args()
    from pytorch_geometric_signed_directed_amd import graphs
    dev = torch.device("cuda:0")
    args.edge_weight = None
    if args.signed:
        ei_np, sign_np, _, _ = graphs.sdsbm_for_edges(args.nodes, args.edges, seed=1)
        ei, args.edge_weight = torch.from_numpy(ei_np).to(dev), torch.from_numpy(sign_np).to(dev)
    else:
        ei = torch.from_numpy(graphs.dsbm_for_edges(args.nodes, args.edges, seed=0)[0]).to(dev)
    g = torch.Generator().manual_seed(0)
    x_real = torch.randn(args.nodes, args.hidden, generator=g).to(dev)
    x_imag = torch.randn(args.nodes, args.hidden, generator=g).to(dev)
    shapes = [("grid", 1, 1), ("grid", 1, 2), ("grid", 1, 4), ("grid", 2, 2), ("grid", 2, 4), ("rows", 1, 1), ("rows", 2, 1)]
    if args.world <= 2:
        shapes = [("rows", 1, 1), ("rows", 2, 1), ("rows", 4, 1)]
    if args.shapes:
        from pytorch_geometric_signed_directed_amd.parallel import split_spec

        def spec(raw):                                   # "2" -> 2 equal pieces, "0.4,0.6" -> uneven ones
            count, fracs = split_spec(raw)
            return fracs if fracs is not None else count
        shapes = [(t.split(":")[0], spec(t.split(":")[1]), spec(t.split(":")[2])) for t in args.shapes]
    out = {"world": args.world, "rank": args.rank, "nodes": args.nodes, "edges": int(ei.size(1)), "hidden": args.hidden,
           "K": args.K, "signed": args.signed, "latency_us": args.latency_us, "single_gpu_ms": args.single_gpu_ms, "runs": []}
    for gbps in args.link_gbps:
        for layout, phases, chunks in shapes:
            try:
                rec = run_shape(args, ei, x_real, x_imag, layout, phases, chunks, gbps)
            except Exception as exc:  # noqa: BLE001  (a shape that does not divide: report and go on)
                rec = {"layout": layout, "phases": phases, "return_chunks": chunks, "link_gbps": gbps, "error": repr(exc)}
            if args.single_gpu_ms and "step_ms_median" in rec:
                rec["projected_speedup"] = args.single_gpu_ms / rec["step_ms_median"]
                if rec.get("step_ms_hipgraph_replay"):
                    rec["projected_speedup_hipgraph_replay"] = args.single_gpu_ms / rec["step_ms_hipgraph_replay"]
            out["runs"].append(rec)
            print(json.dumps(rec), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
