"""How much does one hub node cost the operator BUILDS (sequential per-row degree sums, reference summation order)?
Magnetic Laplacian build and gcn_norm on 100k nodes / 4M edges with one node of 0 / 100k / 1M incident edges."""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_geometric_signed_directed_amd.utils._laplacian import laplacian_parts  # noqa: E402
from pytorch_geometric_signed_directed_amd.utils._norm import gcn_norm  # noqa: E402


def timed(fn, reps=5):
    ts = []
    for k in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        if k:
            ts.append(a.elapsed_time(b))
    return statistics.median(ts)


out = {}
dev = torch.device("cuda:0")
n, e = 100000, 4000000
for hub in (0, 100000, 1000000):
    g = torch.Generator().manual_seed(1)
    ei = torch.randint(0, n, (2, e), generator=g)
    if hub:
        ei[1, :hub] = 17
    ei = ei.to(dev)
    w = torch.rand(e, device=dev) + 0.5
    out[f"hub_{hub}"] = {"laplacian_parts_ms": timed(lambda: laplacian_parts(ei, w, n, False, True)),
                         "gcn_norm_ms": timed(lambda: gcn_norm(ei, w, n))}
    print(hub, out[f"hub_{hub}"], flush=True)
with open(os.path.join(ROOT, "gpurun_out", "hub_build_probe.json"), "w") as fh:
    json.dump(out, fh, indent=1)
