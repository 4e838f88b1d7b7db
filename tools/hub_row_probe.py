"""Measurement helper: cost of one hub row (power-law tail) in the SpMM.  Before the segmented long-row
path (ABI v3) one wavefront owned the hub: 0.141 / 0.302 / 2.74 / 27.1 ms for 0 / 10k / 100k / 1M entries;
with it: 0.141 / 0.176 / 0.182 / 0.240 ms (MI355X)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from pytorch_geometric_signed_directed_amd.sparse import Pattern, _spmm_raw
dev = torch.device("cuda:0")
n, nnz, f = 100000, 4000000, 64
g = torch.Generator(device="cuda").manual_seed(0)
for hub in (0, 10000, 100000, 1000000):
    src = torch.randint(0, n, (nnz + hub,), device=dev, generator=g)
    dst = torch.randint(0, n, (nnz + hub,), device=dev, generator=g)
    if hub:
        dst[:hub] = 7
    pat = Pattern(torch.stack([src, dst]), n, n)
    x = torch.randn(n, f, device=dev)
    w = torch.rand(nnz + hub, device=dev)
    v = pat.values_for(w, "fwd")
    for _ in range(3): _spmm_raw(pat.fwd, v, x, None, 1.0, 0.0, False)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): _spmm_raw(pat.fwd, v, x, None, 1.0, 0.0, False)
    b.record(); torch.cuda.synchronize()
    print(f"hub row of {hub:>8} entries: {a.elapsed_time(b) / 10:.3f} ms per SpMM")
