"""Measurement helper: the dual SpMM over the WHOLE north-star operator at narrow feature widths -- the per-GPU
product of a feature-parallel (column-sharded) multi-GPU layout, where every GPU walks all entries but only
F / n_gpus columns."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_signed_directed_amd.sparse import Pattern
from tools.colblock_probe_lib import run, timeit, dev

g = torch.Generator(device="cuda").manual_seed(0)
n, nnz = 1000000, 41000000
ei = torch.randint(0, n, (2, nnz), device=dev, generator=g)
csr = Pattern(ei, n, n).fwd
va, vb = torch.rand(nnz, device=dev), torch.rand(nnz, device=dev)
for f in (64, 32, 16, 8, 4):
    xa, xb = torch.randn(n, f, device=dev), torch.randn(n, f, device=dev)
    ya, yb = torch.empty_like(xa), torch.empty_like(xb)
    res = []
    for hint in (1, 0):
        res.append(timeit(lambda: run(csr, va, vb, xa, xb, ya, yb, 0, f, f, hint)))
    packed = torch.randn(n, 2 * f, device=dev)                     # [x_real | x_imag] per node: one line per gather
    outp = torch.empty(n, 2 * f, device=dev)
    tp = timeit(lambda: run(csr, va, vb, packed[:, :f], packed[:, f:], outp[:, :f], outp[:, f:], 0, f, 2 * f, 1))
    print(f"F={f:>3}: dual SpMM over 1M rows / 41M entries: light {res[0]:.3f} ms, deep {res[1]:.3f} ms, "
          f"packed operands {tp:.3f} ms", flush=True)
