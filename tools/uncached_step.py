"""Measurement helper: MagNetConv north-star step with cached=False (reference default)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from pytorch_geometric_signed_directed_amd import graphs
from pytorch_geometric_signed_directed_amd.nn import MagNetConv
dev = torch.device("cuda:0")
n, e, h = 1000000, 20000000, 64
ei = torch.from_numpy(graphs.dsbm_for_edges(n, e, seed=0)[0]).to(dev)
g = torch.Generator().manual_seed(0)
xr = torch.randn(n, h, generator=g).to(dev).requires_grad_()
xi = torch.randn(n, h, generator=g).to(dev).requires_grad_()
torch.manual_seed(0)
layer = MagNetConv(h, h, 1, 0.25, False, cached=False).to(dev)
from pytorch_geometric_signed_directed_amd import memo
def step(rebuild):
    layer.zero_grad(set_to_none=True); xr.grad = xi.grad = None
    if rebuild:                       # drop every memoised operator: this forward rebuilds it, as the reference's does
        memo.clear_all()
    o = layer(xr, xi, ei); (o[0].sum() + o[1].sum()).backward()
for rebuild in (True, False):
    for _ in range(2): step(rebuild)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step(rebuild)
    torch.cuda.synchronize()
    print("cached=False,", "operator rebuilt every step" if rebuild else "same graph tensors (memo hit)",
          "ms/step", (time.perf_counter() - t0) / 5 * 1e3)
