"""Measurement helper (GPU box): the other BASELINE.json configs on ONE MI355X -- they are parity-test
sizes, not bench.py lines, but each gets a timing and the achieved algorithmic GB/s of its SpMM kernel
(SURVEY.md 8(d) byte model) so DESIGN.md can quote them.  Writes gpurun_out/configs.json."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
from pytorch_geometric_signed_directed_amd import _cabi, graphs  # noqa: E402
from pytorch_geometric_signed_directed_amd.nn import (Conv_Base, DiGCN_InceptionBlock, MagNetConv, MSConv,  # noqa: E402
                                                      SGCNConv, SIMPA)

dev = torch.device("cuda:0")
out = {}
# e.g. PYGSD_CONFIGS=C3,C5 -- or one step alone for a profile whose kernel averages belong to ONE configuration:
# C3a (SGCNConv), C3b (SIMPA), C5a (inception block fp32), C5b (bf16)
ONLY = [t for t in os.environ.get("PYGSD_CONFIGS", "").split(",") if t]
# PYGSD_CONFIGS_COMPACT=1 (bench.py's `configs` leg): the eager fwd+bwd step of every configuration and its kernel classes only --
# no cached=False variants, no hipGraph replays, no with-loss variant
COMPACT = os.environ.get("PYGSD_CONFIGS_COMPACT", "0") == "1"
OUT_PATH = os.environ.get("PYGSD_CONFIGS_OUT")


def want(tag):
    return not ONLY or tag in ONLY or tag[:2] in ONLY




def timed(step, iters=10, warm=3):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    _cabi.prof_reset(); _cabi.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    _cabi.prof_enable(False)
    prof = {k: _cabi.prof_collect(k) for k in ("spmm", "spmm2", "dense", "dense_bwd", "build", "elementwise")}
    _cabi.prof_reset()
    return ms, {k: {"launches_per_step": n / iters, "ms_per_launch": (t / n if n else 0.0)} for k, (n, t) in prof.items()}


def replayed(step, iters=20):
    """The same step captured into a hipGraph (hipgraph.capture_step) and replayed: the step without its host launch path
    (a few dozen launches of tens of microseconds each).  None when the capture fails."""
    if COMPACT:
        return None
    from pytorch_geometric_signed_directed_amd.hipgraph import capture_step
    try:
        g = capture_step(step)
        for _ in range(3):
            g()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            g()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3
    except Exception as exc:  # noqa: BLE001
        print("hipGraph capture failed:", repr(exc)[:200], flush=True)
        return None


def spmm_bytes(nnz, n, f, s=4, val=True):
    return nnz * (4 + (4 if val else 0) + f * s) + n * f * s + 4 * (n + 1)


def residency(gathered_bytes, gbps):
    """Every algorithmic rate is quoted with the size of the gathered feature set: at or below the 256 MiB Infinity
    Cache the gathers are served on-die, and the byte model (which credits no reuse) can exceed the 8 TB/s HBM peak."""
    mib = gathered_bytes / 2 ** 20
    rec = {"gathered_set_MiB": round(mib, 1), "fraction_of_8TBps": (gbps / 8000.0) if gbps else None}
    if mib <= 256:
        rec["note"] = ("gathered set fits the 256 MiB Infinity Cache: the algorithmic rate is an on-die (L2-miss / "
                       "fabric) rate, NOT a DRAM rate, and may exceed the HBM peak")
    else:
        rec["note"] = "gathered set exceeds the Infinity Cache: served by Infinity Cache + HBM together"
    return rec


def magnetic(name, cls, n, e, h, K, signed, **kw):
    w = None
    if signed:      # SDSBM (data/general/SDSBM.py:10-67) on the signed cyclic meta-graph, 10 % of the signs flipped
        ei_np, sign_np, _, _ = graphs.sdsbm_for_edges(n, e, seed=1)
        w = torch.from_numpy(sign_np).to(dev)
    else:
        ei_np, _, _ = graphs.dsbm_for_edges(n, e, seed=1)
    ei = torch.from_numpy(ei_np).to(dev)
    g = torch.Generator().manual_seed(0)
    xr = torch.randn(n, h, generator=g).to(dev).requires_grad_()
    xi = torch.randn(n, h, generator=g).to(dev).requires_grad_()
    torch.manual_seed(0)
    layer = cls(h, h, K, 0.25, False, cached=True, **kw).to(dev)

    def step():
        layer.zero_grad(set_to_none=True); xr.grad = xi.grad = None
        o = layer(xr, xi, ei, w)
        (o[0].sum() + o[1].sum()).backward()
    ms, prof = timed(step)
    nnz = layer._operator.nnz
    b = spmm_bytes(nnz, n, h) + spmm_bytes(nnz - n, n, h)
    k = prof["spmm2"]
    out[name] = {"nodes": n, "edges": int(ei.size(1)), "hidden": h, "K": K, "ms_per_step": ms,
                 "edges_per_s": ei.size(1) / ms * 1e3, "operator_nnz": nnz, "kernels": prof,
                 # one logical product may be several column-block launches (F >= 128): price it as a whole
                 "spmm2_alg_GBps": (b / (k["ms_per_launch"] * k["launches_per_step"] / (2 * K)) / 1e6
                                    if k["ms_per_launch"] else None)}
    out[name].update(residency(2 * n * h * 4, out[name]["spmm2_alg_GBps"]))
    if COMPACT:
        print(name, json.dumps(out[name]), flush=True)
        del layer, xr, xi, ei
        torch.cuda.empty_cache()
        return
    # reference default: cached=False, operator rebuilt on every forward
    layer_u = cls(h, h, K, 0.25, False, cached=False, **kw).to(dev)
    layer_u.load_state_dict(layer.state_dict())

    def step_u(rebuild=True):
        layer_u.zero_grad(set_to_none=True); xr.grad = xi.grad = None
        if rebuild:          # as if a new graph tensor arrived: the layer's operator memo cannot hit
            layer_u._op_memo.clear(); layer_u._parts_memo.clear()
        o = layer_u(xr, xi, ei, w)
        (o[0].sum() + o[1].sum()).backward()
    ms_u, prof_u = timed(step_u, iters=5, warm=2)
    out[name]["uncached_ms_per_step"] = ms_u
    out[name]["uncached_build_ms"] = prof_u["build"]["launches_per_step"] * prof_u["build"]["ms_per_launch"]
    ms_m, _ = timed(lambda: step_u(False), iters=5, warm=2)
    out[name]["uncached_same_tensors_ms_per_step"] = ms_m      # cached=False, unchanged graph tensors
    print(name, json.dumps(out[name]), flush=True)
    del layer, layer_u, xr, xi, ei
    torch.cuda.empty_cache()


def signed_c3(n=500000, entries=10000000, h=64):
    p = (entries / 2) / (n * (n - 1) / 2)
    ei_np, sign, _ = graphs.ssbm(n, 5, p, 0.1, 2.0, seed=2)
    ei, sign = torch.from_numpy(ei_np).to(dev), torch.from_numpy(sign).to(dev)
    pos, neg = ei[:, sign > 0].contiguous(), ei[:, sign < 0].contiguous()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, h, generator=g).to(dev).requires_grad_()
    torch.manual_seed(0)
    conv = SGCNConv(h, h // 2, first_aggr=True).to(dev)

    # PYGSD_DENSE_GRAD=1: the upstream gradient is a dense [N, h] matrix handed to backward (what a real loss delivers), instead of
    # `.sum()` -- whose reduction, ones-fill and materialised broadcast gradient are torch kernels timed with the layer
    g_up = torch.randn(n, h, generator=g).to(dev)
    dense_grad = os.environ.get("PYGSD_DENSE_GRAD", "0") == "1"

    def step():
        conv.zero_grad(set_to_none=True); x.grad = None
        if dense_grad:
            torch.autograd.backward(conv(x, pos, neg), g_up)
        else:
            conv(x, pos, neg).sum().backward()
    if not want("C3a"):
        ms, prof = None, None
    else:
        ms, prof = timed(step)
    # the layer applies its Linear blocks BEFORE the aggregation, so the value-less mean SpMMs run at width h / 2
    if prof is not None:
        b = spmm_bytes(pos.size(1), n, h // 2, val=False) + spmm_bytes(neg.size(1), n, h // 2, val=False)
        k = prof["spmm"]
        out["C3_sgcnconv_first"] = {"nodes": n, "pos_entries": int(pos.size(1)), "neg_entries": int(neg.size(1)),
                                    "hidden": h, "ms_per_step": ms, "ms_per_step_hipgraph_replay": replayed(step),
                                    "upstream_gradient": "dense [N, h] handed to backward" if dense_grad else "out.sum()",
                                    "entries_per_s": ei.size(1) / ms * 1e3, "kernels": prof,
                                    "spmm_alg_GBps_fwd_pair": b / (2 * k["ms_per_launch"]) / 1e6 if k["ms_per_launch"] else None}
        out["C3_sgcnconv_first"].update(residency(n * (h // 2) * 4, out["C3_sgcnconv_first"]["spmm_alg_GBps_fwd_pair"]))
        print("C3_sgcnconv_first", json.dumps(out["C3_sgcnconv_first"]), flush=True)
    if not want("C3b"):
        torch.cuda.empty_cache()
        return
    simpa = SIMPA(2, 0.5).to(dev)
    wp = torch.ones(pos.size(1), device=dev)
    wn = torch.ones(neg.size(1), device=dev)
    xp = torch.randn(n, h, generator=g).to(dev).requires_grad_()
    xn = torch.randn(n, h, generator=g).to(dev).requires_grad_()

    def step2():
        simpa.zero_grad(set_to_none=True); xp.grad = xn.grad = None
        simpa(pos, wp, neg, wn, xp, xn).sum().backward()
    ms, prof = timed(step2)
    # 6 products forward + their 6 transposes backward, 4 on A_p and 2 on A_n each way, all at width h with values
    k = prof["spmm"]
    b_step = 2 * (4 * spmm_bytes(pos.size(1) + n, n, h) + 2 * spmm_bytes(neg.size(1) + n, n, h))
    spmm_ms = k["ms_per_launch"] * k["launches_per_step"]
    out["C3_simpa_hop2"] = {"nodes": n, "hidden": h, "pos_entries": int(pos.size(1)), "neg_entries": int(neg.size(1)), "ms_per_step": ms,
                            "spmm_alg_GBps": b_step / spmm_ms / 1e6 if spmm_ms else None,
                            "ms_per_step_hipgraph_replay": replayed(step2), "entries_per_s": ei.size(1) / ms * 1e3,
                            "kernels": prof, "note": "6 SpMM fwd + 6 bwd (4 on A_p, 2 on A_n) per step; the reference's unused last-hop product is skipped"}
    out["C3_simpa_hop2"].update(residency(n * h * 4, out["C3_simpa_hop2"]["spmm_alg_GBps"]))
    print("C3_simpa_hop2", json.dumps(out["C3_simpa_hop2"]), flush=True)
    torch.cuda.empty_cache()


def digcn_c5(n=2000000, e=25000000, h=64):
    ei_np, _, _ = graphs.dsbm_for_edges(n, e, seed=3)
    src, dst = torch.from_numpy(ei_np).to(dev)
    loops = torch.arange(n, device=dev)
    res = {}
    for dtype in (torch.float32, torch.bfloat16):
        if not want("C5a" if dtype == torch.float32 else "C5b"):
            continue
        ops = []
        for k in range(2):           # two symmetric, positively weighted, sym-normalised operators
            g = torch.Generator(device="cuda").manual_seed(10 + k)
            perm = torch.randperm(src.numel(), device=dev, generator=g) if k else torch.arange(src.numel(), device=dev)
            s, d = src[perm], (dst if k == 0 else dst[torch.randperm(dst.numel(), device=dev, generator=g)])
            wv = torch.rand(s.numel(), device=dev, generator=g)
            ei = torch.stack([torch.cat([s, d, loops]), torch.cat([d, s, loops])])
            w = torch.cat([wv, wv, torch.ones(n, device=dev)])
            deg = torch.zeros(n, device=dev).index_add_(0, ei[0], w)
            w = deg[ei[0]].rsqrt() * w * deg[ei[1]].rsqrt()
            ops.append((ei, w))
        x = torch.randn(n, h, device=dev).to(dtype).requires_grad_()
        torch.manual_seed(0)
        ib = DiGCN_InceptionBlock(h, h).to(dev).to(dtype)

        # the loss is NOT part of the block: three pre-built upstream gradients (one row broadcast over the nodes each, as a
        # sum-type loss would hand them over) go straight into autograd -- the round-3 figure included 0.58 ms of
        # (x0 + x1 + x2).float().sum() and its materialised gradient, reported separately below
        grow = [torch.ones(1, h, device=dev, dtype=dtype).expand(n, h) for _ in range(3)]

        def step():
            ib.zero_grad(set_to_none=True); x.grad = None
            outs = ib(x, ops[0][0], ops[0][1], ops[1][0], ops[1][1])
            torch.autograd.backward(outs, grow)

        def step_with_loss():
            ib.zero_grad(set_to_none=True); x.grad = None
            x0, x1, x2 = ib(x, ops[0][0], ops[0][1], ops[1][0], ops[1][1])
            (x0 + x1 + x2).float().sum().backward()
        ms_loss = None if COMPACT else timed(step_with_loss, iters=5, warm=2)[0]
        ms, prof = timed(step, iters=5, warm=2)
        nnz = ops[0][0].size(1)
        s_el = 2 if dtype == torch.bfloat16 else 4
        b = spmm_bytes(nnz, n, h, s=s_el)
        k = prof["spmm"]
        res[str(dtype).split(".")[-1]] = {"ms_per_block_step": ms, "ms_per_block_step_incl_sum_loss": ms_loss,
                                          "ms_per_block_step_hipgraph_replay": replayed(step),
                                          "nnz_per_operator": nnz, "kernels": prof,
                                          "spmm_alg_GBps": b / k["ms_per_launch"] / 1e6 if k["ms_per_launch"] else None,
                                          "nnz_per_s": 2 * nnz / ms * 1e3}
        res[str(dtype).split(".")[-1]].update(residency(n * h * s_el, res[str(dtype).split(".")[-1]]["spmm_alg_GBps"]))
        del ops, x, ib
        torch.cuda.empty_cache()
    out["C5_digcn_inception_block_1gpu"] = {"nodes": n, "hidden": h, **res}
    print("C5", json.dumps(out["C5_digcn_inception_block_1gpu"]), flush=True)


if want("C2"):
    magnetic("C2_magnetconv_100k_2M_h64", MagNetConv, 100000, 2000000, 64, 1, False)
if want("C3") or want("C3a") or want("C3b"):
    signed_c3()
if want("C4"):
    magnetic("C4_msconv_1M_20M_h128_K2_1gpu", MSConv, 1000000, 20000000, 128, 2, True)
if want("northstar"):
    magnetic("northstar_magnetconv_1M_20M_h64", MagNetConv, 1000000, 20000000, 64, 1, False)
if want("C5") or want("C5a") or want("C5b"):
    digcn_c5()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(OUT_PATH or ("gpurun_out/configs.json" if not ONLY else "gpurun_out/configs_partial.json"), "w"), indent=1)
