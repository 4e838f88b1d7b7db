"""Measurement helper (GPU box): the `cached=False` operator build at the
north-star size -- fused pipeline (csrc/magop.hip) against the generic one (csrc/laplacian.hip), unweighted
(MagNetConv, the bench graph) and weighted / signed (MSConv, BASELINE config C4's signs).  HIP-event time per
build, median of `--iters`.  Run under `rocprofv3 --kernel-trace --stats` (and `--pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` passes) for the per-kernel table committed as profiles/r3_build_kernel_stats.csv.
Writes gpurun_out/build_probe.json."""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_signed_directed_amd import graphs  # noqa: E402
from pytorch_geometric_signed_directed_amd.utils._laplacian import (assemble_operator_csr, fused_operator_csr,  # noqa: E402
                                                                     laplacian_parts, laplacian_values, set_unit_build)

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=1000000)
ap.add_argument("--edges", type=int, default=20000000)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--norm", default="sym", help="sym | none")
ap.add_argument("--only", default="", help="comma list of legs: fused (= round 4's one-pass unweighted build), two_stage (round 3's), "
                                           "generic, fused_signed (+-1 weights: round 5's one-call build), two_stage_signed, "
                                           "fused_real_weights (round 5: the weighted bucket form), sorted_real_weights (behind the radix sort), "
                                           "generic_signed")
args = ap.parse_args()
dev = torch.device("cuda:0")
NORM = None if args.norm == "none" else "sym"


def generic(ei, w, n, signed):
    parts = laplacian_parts(ei, w, n, signed, True)
    off_r, off_i, diag, mir_r, mir_i = laplacian_values(parts, 0.25, NORM, mirror=True)
    return assemble_operator_csr(parts, off_r, off_i, mir_r, mir_i, diag, 2.0, -1.0)


def fused(ei, w, n, signed):
    return fused_operator_csr(ei, w, n, signed, True, 0.25, NORM, 2.0)


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return {"median_ms": statistics.median(ts), "min_ms": min(ts), "max_ms": max(ts)}


n, e = args.nodes, args.edges
ei_np, _, _ = graphs.dsbm_for_edges(n, e, seed=0)
ei = torch.from_numpy(ei_np).to(dev)
ei_s_np, sign_np, _, _ = graphs.sdsbm_for_edges(n, e, seed=1)
ei_s, w_s = torch.from_numpy(ei_s_np).to(dev), torch.from_numpy(sign_np).to(dev)
def two_stage():
    prev = set_unit_build(False)
    try:
        return fused(ei, None, n, False)
    finally:
        set_unit_build(prev)


def two_stage_signed():
    from pytorch_geometric_signed_directed_amd.utils._laplacian import set_signed_unit_build
    prev = set_signed_unit_build(False)
    try:
        return fused(ei_s, w_s, n, True)
    finally:
        set_signed_unit_build(prev)


g_w = torch.Generator(device=dev).manual_seed(3)
w_real = w_s * (torch.rand(w_s.shape, generator=g_w, device=dev) + 0.5)     # real-valued signed weights: the two-stage pipeline
def sorted_real_weights():
    os.environ["PYGSD_WEIGHTED_BUILD_FORM"] = "sort"
    try:
        return fused(ei_s, w_real, n, True)
    finally:
        os.environ.pop("PYGSD_WEIGHTED_BUILD_FORM", None)


legs = {"fused": lambda: fused(ei, None, n, False), "two_stage": two_stage, "generic": lambda: generic(ei, None, n, False),
        "sorted_real_weights": sorted_real_weights,
        "fused_signed": lambda: fused(ei_s, w_s, n, True), "two_stage_signed": two_stage_signed,
        "fused_real_weights": lambda: fused(ei_s, w_real, n, True), "generic_signed": lambda: generic(ei_s, w_s, n, True)}
only = [s for s in args.only.split(",") if s] or list(legs)
out = {"nodes": n, "edges": int(ei.size(1)), "iters": args.iters}
csr = fused(ei, None, n, False)[0]
out["operator_nnz"] = csr.nnz
# compulsory bytes of one build: int64 COO in; one int32 CSR + four fp32 value arrays out
out["compulsory_bytes"] = 16 * int(ei.size(1)) + 4 * (n + 1) + 20 * csr.nnz
for name in only:
    out[name] = timed(legs[name], args.iters)
    out[name]["compulsory_GBps"] = out["compulsory_bytes"] / out[name]["median_ms"] / 1e6
    print(name, json.dumps(out[name]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/build_probe.json", "w"), indent=1)
