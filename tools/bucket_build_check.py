"""Measurement / check helper (GPU box): the unweighted operator build in its bucket form (csrc/magop.hip, bucket_pass /
bucket_merge_rows) against the radix-sort form (PYGSD_UNIT_BUILD_FORM=sort) -- bit equality of every output array on a few graphs,
then HIP-event time of both at the north-star size.  Writes gpurun_out/bucket_build_check.json."""
import json
import os
import statistics
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_signed_directed_amd import graphs  # noqa: E402
from pytorch_geometric_signed_directed_amd.utils._laplacian import fused_operator_csr  # noqa: E402

dev = torch.device("cuda:0")


def build(ei, n, form, norm="sym"):
    if form:
        os.environ["PYGSD_UNIT_BUILD_FORM"] = form
    else:
        os.environ.pop("PYGSD_UNIT_BUILD_FORM", None)
    try:
        return fused_operator_csr(ei, None, n, False, True, 0.25, norm, 2.0)
    finally:
        os.environ.pop("PYGSD_UNIT_BUILD_FORM", None)


def same(a, b):
    ca, fa, ba, da = a
    cb, fb, bb, db = b
    ok = ca.nnz == cb.nnz and torch.equal(ca.rowptr, cb.rowptr) and torch.equal(ca.col, cb.col) and torch.equal(da, db)
    for x, y in zip(fa + ba, fb + bb):
        ok = ok and torch.equal(x.view(torch.int32), y.view(torch.int32))
    return bool(ok)


def timed(fn, iters=10):
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
    return {"median_ms": statistics.median(ts), "min_ms": min(ts)}


out = {"equal": {}}
rng = np.random.default_rng(0)
cases = {}
for name, n, e in (("tiny", 37, 200), ("small", 2000, 30000), ("mid", 100000, 2000000)):
    r = rng.integers(0, n, e)
    c = rng.integers(0, n, e)
    ei = np.stack([r, c])
    ei = np.concatenate([ei, ei[:, : e // 10], ei[::-1, e // 10: e // 5]], axis=1)      # duplicates, reciprocal pairs (+ self loops)
    cases[name] = (torch.from_numpy(ei).to(dev), n)
hub = np.stack([np.concatenate([np.zeros(300, np.int64), rng.integers(0, 5000, 20000)]),
                np.concatenate([np.arange(1, 301), rng.integers(0, 5000, 20000)])])
cases["hub300"] = (torch.from_numpy(hub).to(dev), 5000)
ei_np, _, _ = graphs.dsbm_for_edges(1000000, 20000000, seed=0)
cases["north_star"] = (torch.from_numpy(ei_np).to(dev), 1000000)
for name, (ei, n) in cases.items():
    for norm in ("sym", None):
        a, b = build(ei, n, None, norm), build(ei, n, "sort", norm)
        out["equal"][f"{name}_{norm}"] = same(a, b)
        print(name, norm, out["equal"][f"{name}_{norm}"], a[0].nnz, flush=True)
    a2 = build(ei, n, None)
    out["equal"][f"{name}_deterministic"] = same(build(ei, n, None), a2)
ei, n = cases["north_star"]
for form in ("bucket", "sort"):
    out[form] = timed(lambda: build(ei, n, None if form == "bucket" else "sort"))
    print(form, out[form], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/bucket_build_check.json", "w"), indent=1)
