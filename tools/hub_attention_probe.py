"""Hub-row cliff of the attention kernels, before / after the segment-parallel path: GAT aggregate fwd + bwd
(nn/signed/GATConv._GatAggregate) on 100k rows / 4M entries at F = 64 with one row of 0 / 10k / 100k / 1M entries.
PYGSD_DISABLE_SEGMENT_HUBS=1 (probe-only switch in this script) hides the hub list from the attention kernels."""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib  # noqa: E402

import pytorch_geometric_signed_directed_amd.sparse as S  # noqa: E402

G = importlib.import_module("pytorch_geometric_signed_directed_amd.nn.signed.GATConv")   # the module, not the class


def run(hub, use_hubs):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    n, e, f = 100000, 4000000, 64
    ei = torch.randint(0, n, (2, e), generator=g)
    if hub:
        ei[1, :hub] = 17
    pat = S.Pattern(ei.to(dev), n, n)
    h = torch.randn(n, f, generator=g).to(dev).requires_grad_()
    a_s, a_d = torch.randn(n, generator=g).to(dev).requires_grad_(), torch.randn(n, generator=g).to(dev).requires_grad_()
    real = S.segment_long_rows_arg
    if not use_hubs:
        G.segment_long_rows_arg = lambda csr: (None, None)
    try:
        def step():
            out = G._GatAggregate.apply(h, a_s, a_d, pat, 0.2)
            out.sum().backward()
        for _ in range(2):
            step()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); step(); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
    finally:
        G.segment_long_rows_arg = real
    return statistics.median(ts)


def main():
    out = {}
    for hub in (0, 10000, 100000, 1000000):
        out[f"hub_{hub}"] = {"segment_path_ms": run(hub, True), "row_per_wavefront_ms": run(hub, False)}
        print(hub, out[f"hub_{hub}"], flush=True)
    with open(os.path.join(ROOT, "gpurun_out", "hub_attention_probe.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
