"""Measurement helper (GPU box): achieved algorithmic GB/s of the fused dual SpMM vs working-set size
(does the 256 MiB Infinity Cache serve gathers faster than HBM?)."""
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
from pytorch_geometric_signed_directed_amd.sparse import Pattern, _spmm2_raw, _spmm_raw
dev = torch.device("cuda:0")
f, deg = 64, 41
res = []
for n in (62500, 125000, 250000, 500000, 1000000, 2000000):
    nnz = n * deg
    g = torch.Generator(device="cuda").manual_seed(0)
    ei = torch.stack([torch.randint(0, n, (nnz,), device=dev, generator=g), torch.randint(0, n, (nnz,), device=dev, generator=g)])
    pat = Pattern(ei, n, n)
    va, vb = torch.rand(nnz, device=dev), torch.rand(nnz, device=dev)
    xa, xb = torch.randn(n, f, device=dev), torch.randn(n, f, device=dev)
    for name, fn, nb in (("spmm2", lambda: _spmm2_raw(pat.fwd, va, vb, xa, xb, None, None, 1.0, 0.0), 2),
                         ("spmm1", lambda: _spmm_raw(pat.fwd, va, xa, None, 1.0, 0.0, False), 1)):
        for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        byt = nnz * (4 + 4 * nb + f * 4 * nb) + nb * n * f * 4 + 4 * (n + 1)
        res.append({"kernel": name, "n": n, "x_bytes_MiB": nb * n * f * 4 / 2**20, "ms": ms, "alg_GBps": byt / ms / 1e6})
        print(res[-1], flush=True)
json.dump(res, open("gpurun_out/spmm_sweep.json", "w"), indent=1)
