"""Measurement helper: does column-blocking a wide SpMM (F=128 -> 2 x 64 columns, F=64 -> 2 x 32) pay once the
gathered set exceeds the 256 MB Infinity Cache?  Launches the C-ABI directly with strided views."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from pytorch_geometric_signed_directed_amd.sparse import Pattern

from tools.colblock_probe_lib import run, timeit, dev

g = torch.Generator(device="cuda").manual_seed(0)
for n, nnz, f, dual in ((1000000, 41000000, 128, True), (2000000, 52000000, 64, False), (1000000, 41000000, 64, True),
                        (2000000, 52000000, 128, False)):
    ei = torch.randint(0, n, (2, nnz), device=dev, generator=g)
    pat = Pattern(ei, n, n)
    csr = pat.fwd
    va = torch.rand(nnz, device=dev)
    vb = torch.rand(nnz, device=dev) if dual else None
    xa, xb = torch.randn(n, f, device=dev), torch.randn(n, f, device=dev)
    ya, yb = torch.empty_like(xa), torch.empty_like(xb)
    t1 = timeit(lambda: run(csr, va, vb, xa, xb, ya, yb, 0, f, f))
    ref = ya.clone()
    res = {"1": t1}
    for parts in (2, 4):
        w = f // parts
        t = timeit(lambda: [run(csr, va, vb, xa, xb, ya, yb, i * w, w, f) for i in range(parts)])
        assert torch.allclose(ref, ya, atol=1e-3, rtol=1e-4)
        res[str(parts)] = t
    for parts in (1, 2):
        w = f // parts
        for hint, name in ((1, "light"), (0, "deep")):
            res[f"{parts}/{name}"] = timeit(lambda: [run(csr, va, vb, xa, xb, ya, yb, i * w, w, f, hint) for i in range(parts)])
    print(f"n={n} nnz={nnz} F={f} dual={dual}: " + "  ".join(f"{k} block(s) {v:.3f} ms" for k, v in res.items()), flush=True)
    del pat, csr, ei
