"""bench.py -- edges/sec (fwd+bwd) of MagNetConv on a synthetic DSBM graph, one JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nodes 1000000] [--edges 20000000]
                    [--hidden 64] [--no-cpu-baseline]

Metric (BASELINE.json): input edges per second, forward + backward of ONE MagNetConv layer
(K=1, q=0.25, sym normalisation, in = out = hidden, fp32), operator cached (steady state), on a DSBM
graph (5 clusters, cyclic meta-graph eta=0.1, size_ratio 1.5) of 1M nodes / 20M edges; inputs
resident in HBM before the timed region.  A "step" = layer forward + loss.backward() with
loss = out_real.sum() + out_imag.sum() and gradients w.r.t. x_real, x_imag, weight, bias.

N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL): the same graph is sharded by
node range across the ranks (strong scaling; `parallel.ShardedMagNetConv`).  Default work layout: a
p_r x p_c process grid (row blocks of the operator x column slices of the features) with two
all-to-alls per propagate over xGMI; `--layout rows` = the plain all-gather of whole feature blocks.

Extra objects in the JSON line: `roofline` (dominant kernel = the fused dual-value SpMM, timed by
HIP events around every launch inside the timed region, algorithmic bytes per SURVEY.md 8(d)) and
`cpu_baseline` (the oracle's reference op sequence timed on this box's host cores, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def spmm_bytes(nnz, n, f, s=4):
    """ALGORITHMIC bytes of one SpMM Y = S X (edge-centric gather model, int32 CSR,
    no cache-reuse credit): nnz * (col 4 + val 4 + gathered row F*s) + N*F*s (write) + rowptr."""
    return nnz * (8 + f * s) + n * f * s + 4 * (n + 1)


def build_inputs(n, e, hidden, device, seed=0):
    from pytorch_geometric_signed_directed_amd import graphs
    ei, _, p = graphs.dsbm_for_edges(n, e, seed=seed)
    g = torch.Generator().manual_seed(0)
    x_real = torch.randn(n, hidden, generator=g)
    x_imag = torch.randn(n, hidden, generator=g)
    edge_index = torch.from_numpy(ei)
    return edge_index.to(device), x_real.to(device), x_imag.to(device), p


def cpu_baseline(hidden, steps=2, n=100000, e=2000000, threads=None):
    """Reference op sequence (index_select -> mul -> scatter_add_, 4 propagates per order incl. the
    reference's duplicates, autograd backward) on the host cores, cached operator."""
    from oracle import ref_layers as R
    from pytorch_geometric_signed_directed_amd import graphs
    # ATen's index_select / scatter_add_ stop scaling long before a 256-thread host is full (measured
    # on the GPU box, EPYC 9575F: 8 thr 3.8 s, 32 thr 3.2 s, 128 thr 4.7 s, 256 thr 28 s per step), so
    # the baseline runs on the best-performing thread count, and reports that count as `cores`.
    cores = threads or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    ei_np, _, _ = graphs.dsbm_for_edges(n, e, seed=0)
    ei = torch.from_numpy(ei_np)
    g = torch.Generator().manual_seed(0)
    xr = torch.randn(n, hidden, generator=g, requires_grad=True)
    xi = torch.randn(n, hidden, generator=g, requires_grad=True)
    torch.manual_seed(0)
    w = torch.empty(2, hidden, hidden).uniform_(-1, 1).mul_((6.0 / (2 * hidden)) ** 0.5).requires_grad_()
    b = torch.zeros(hidden, requires_grad=True)
    op = R.magnet_operator(ei, None, n, 0.25, "sym", 2.0)

    def step():
        o_r, o_i = R.magnet_conv(xr, xi, op, w, b, duplicate=True)
        (o_r.sum() + o_i.sum()).backward()

    step()  # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    return {"value": ei.size(1) / dt, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"oracle/ref_layers.magnet_conv (reference op sequence, 4 propagates/order), DSBM "
                      f"{n} nodes / {ei.size(1)} edges, h={hidden}, cached operator, {steps} fwd+bwd steps "
                      f"({dt:.2f} s/step)"}


def main():
    # Only the JSON line may reach stdout.  RCCL prints a version banner through C stdio on stdout (flushed at
    # exit, i.e. AFTER the result line), so the real stdout is kept aside for the result and fd 1 is pointed at
    # stderr for everything else (C libraries and stray prints alike).
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nodes", type=int, default=1000000)
    ap.add_argument("--edges", type=int, default=20000000)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layout", choices=("auto", "rows", "grid"), default="auto",
                    help="multi-GPU work layout (parallel.ShardedMagNetConv): rows = all-gather of whole feature "
                         "blocks, grid = p_r x p_c process grid with column-slice all-to-alls (auto picks grid)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the node-sharded layer even with one rank (exercises the RCCL path on 1 GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with python -m torch.distributed.run "
                             "--nproc-per-node N (one rank per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    if os.environ.get("PYGSD_BENCH_SHARE_GPU") == "1":   # test hook: all ranks on cuda:0 (needs the gloo backend)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("PYGSD_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from pytorch_geometric_signed_directed_amd import _cabi
    from pytorch_geometric_signed_directed_amd.nn import MagNetConv

    n, hidden = args.nodes, args.hidden
    edge_index, x_real, x_imag, p = build_inputs(n, args.edges, hidden, device)
    e = edge_index.size(1)
    torch.manual_seed(0)
    if not sharded:
        layer = MagNetConv(hidden, hidden, K=1, q=0.25, trainable_q=False, cached=True).to(device)
        x_real.requires_grad_()
        x_imag.requires_grad_()

        def step():
            layer.zero_grad(set_to_none=True)
            x_real.grad = x_imag.grad = None
            o_r, o_i = layer(x_real, x_imag, edge_index)
            (o_r.sum() + o_i.sum()).backward()

        def op_nnz():
            return layer._operator.nnz
    else:
        from pytorch_geometric_signed_directed_amd.parallel import ShardedMagNetConv
        layer = ShardedMagNetConv(hidden, hidden, K=1, q=0.25, num_nodes=n, edge_index=edge_index,
                                  edge_weight=None, device=device, layout=args.layout)
        xr_loc, xi_loc = layer.shard_rows(x_real).requires_grad_(), layer.shard_rows(x_imag).requires_grad_()
        del x_real, x_imag

        def step():
            layer.zero_grad(set_to_none=True)
            xr_loc.grad = xi_loc.grad = None
            o_r, o_i = layer(xr_loc, xi_loc)
            (o_r.sum() + o_i.sum()).backward()

        def op_nnz():
            return layer.global_nnz

    def sync():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    sync()
    _cabi.prof_reset()
    _cabi.prof_enable(True)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    _cabi.prof_enable(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    launches, kernel_ms = _cabi.prof_collect("spmm2")
    other = {k: _cabi.prof_collect(k) for k in ("dense", "dense_bwd")}
    _cabi.prof_reset()

    if rank == 0:
        nnz = op_nnz()                      # E_s + N (folded diagonal)
        e_s = nnz - n
        per_launch_nnz, per_launch_rows, width = nnz, n, hidden
        parallelism = "single GPU"
        if sharded and getattr(layer, "layout", "rows") == "grid":
            # rank (i, j) multiplies row block i (1 / p_r of the entries) by column slice j (hidden / p_c columns)
            p_r, p_c = layer.plan.p_r, layer.plan.p_c
            per_launch_nnz, per_launch_rows, e_s, width = nnz / p_r, n / p_r, e_s / p_r, hidden // p_c
            parallelism = (f"node-range ownership x{world}, {p_r} x {p_c} process grid: column-slice all-to-all in, "
                           f"row-group all-to-all out (RCCL)")
        elif world > 1 or sharded:
            per_launch_nnz, per_launch_rows = nnz / world, n / world
            e_s = e_s / world
            parallelism = f"node-range shards x{world}, RCCL all-gather of features"
        # one fused launch = the real SpMM (E_s + N entries) + the imaginary SpMM (E_s entries)
        alg_bytes = spmm_bytes(per_launch_nnz, per_launch_rows, width) + spmm_bytes(e_s, per_launch_rows, width)
        avg_ms = kernel_ms / max(launches, 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if launches else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            with open(pmc) as fh:
                rec = json.load(fh)
            if rec.get("nodes") == n and rec.get("hidden") == hidden and rec.get("n_gpus", 1) == world:
                traffic = rec.get("hbm_bytes_per_launch")
        line = {
            "metric": "edges/sec (fwd+bwd) MagNetConv, 1M nodes/20M edges, h=64; % HBM roofline",
            "value": e * args.steps / dt,
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"MagNetConv K=1 q=0.25 sym cached, DSBM(5 clusters, cyclic eta=0.1, "
                                   f"size_ratio 1.5, p={p:.3e}) {n} nodes / {e} edges, h={hidden}, fp32",
                       "nodes": n, "edges": e, "hidden": hidden, "operator_nnz": int(nnz),
                       "parallelism": parallelism},
            "roofline": {"bound": "hbm", "kernel": "spmm_vec_kernel<16,true,true> = LPR 16, dual operator, deep pipelining (pygsd_spmm2_csr_f32)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "launches": int(launches), "avg_launch_ms": avg_ms,
                         "algorithmic_bytes_per_launch": alg_bytes},
            "kernel_ms_per_step": {"spmm2": kernel_ms / args.steps,
                                   **{k: v[1] / args.steps for k, v in other.items()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(hidden)
        result_out.write(json.dumps(line) + "\n")
        result_out.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
