"""Measurement helper: deep vs light gather pipelining of spmm_vec_kernel over the average row length
(F=64, ~40M entries, uniform random columns), single and dual operator.  Feeds the nnz_hint threshold."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from pytorch_geometric_signed_directed_amd.sparse import Pattern
from tools.colblock_probe_lib import run, timeit

dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
f = 64
for deg in (12, 16, 20, 24, 28, 32, 40, 48, 64, 96):
    nnz = 40000000
    n = nnz // deg
    ei = torch.randint(0, n, (2, nnz), device=dev, generator=g)
    csr = Pattern(ei, n, n).fwd
    va, vb = torch.rand(nnz, device=dev), torch.rand(nnz, device=dev)
    xa, xb = torch.randn(n, f, device=dev), torch.randn(n, f, device=dev)
    ya, yb = torch.empty_like(xa), torch.empty_like(xb)
    out = []
    for dual in (False, True):
        for hint in (1, 0):
            out.append(timeit(lambda: run(csr, va, vb if dual else None, xa, xb, ya, yb, 0, f, f, hint)))
    print(f"avg row {deg:>3} n={n:>8}: single light {out[0]:.3f} deep {out[1]:.3f} | dual light {out[2]:.3f} deep {out[3]:.3f}",
          flush=True)
    del csr, ei
