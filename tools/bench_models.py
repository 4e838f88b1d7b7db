"""Measurement helper (GPU box): full train steps (forward, NLL, backward, Adam) of model-level callers at the
benchmark graph sizes -- finds host-side or library-kernel bottlenecks around the layers."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_signed_directed_amd import graphs  # noqa: E402
from pytorch_geometric_signed_directed_amd.nn import (DiGCN_node_classification, MagNet_node_classification,  # noqa: E402
                                                      SGCN, SSSNET_node_clustering)

dev = torch.device("cuda:0")
which = sys.argv[1:] or ["magnet", "sssnet", "sgcn", "digcn"]
out = {}


def timed(step, iters=5, warm=2):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


if "magnet" in which:
    n, e, f = 1000000, 20000000, 64
    ei = torch.from_numpy(graphs.dsbm_for_edges(n, e, seed=0)[0]).to(dev)
    x = torch.randn(n, f, device=dev)
    y = torch.randint(0, 5, (n,), device=dev)
    torch.manual_seed(0)
    m = MagNet_node_classification(f, hidden=64, K=1, label_dim=5, layer=2, dropout=0.5, cached=True).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=0.01)

    def step():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.nll_loss(m(x, x, ei), y).backward()
        opt.step()
    out["MagNet_node_classification 2 layers h=64 K=1, 1M / 20M"] = timed(step)
    del m, ei, x, y
if "sssnet" in which or "sgcn" in which:
    n = 500000
    g = torch.Generator(device="cuda").manual_seed(1)
    pos = torch.randint(0, n, (2, 3000000), device=dev, generator=g)
    neg = torch.randint(0, n, (2, 7000000), device=dev, generator=g)
if "sssnet" in which:
    x = torch.randn(n, 64, device=dev)
    torch.manual_seed(0)
    m = SSSNET_node_clustering(64, 64, 5, 0.5, 2, 0.5, False).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=0.01)
    wp, wn = torch.rand(pos.size(1), device=dev), torch.rand(neg.size(1), device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        _, logp, _, _ = m(pos, wp, neg, wn, x)
        logp.sum().backward()
        opt.step()
    out["SSSNET_node_clustering hop 2 h=64, 500k / 10M signed entries"] = timed(step)
    del m
if "sgcn" in which:
    es = torch.cat([torch.cat([pos.t(), torch.ones(pos.size(1), 1, dtype=torch.long, device=dev)], 1),
                    torch.cat([neg.t(), -torch.ones(neg.size(1), 1, dtype=torch.long, device=dev)], 1)])
    torch.manual_seed(0)
    m = SGCN(n, es, in_dim=64, out_dim=64, layer_num=2, init_emb=torch.randn(n, 64, device=dev)).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=0.01)

    def step():
        opt.zero_grad(set_to_none=True)
        m.loss().backward()
        opt.step()
    out["SGCN 2 layers 64 -> 64 incl. its sampled objectives, 500k / 10M signed entries"] = timed(step)
    del m
if "digcn" in which:
    n, e = 2000000, 50000000
    g = torch.Generator(device="cuda").manual_seed(2)
    ei = torch.randint(0, n, (2, e), device=dev, generator=g)
    w = torch.rand(e, device=dev) / 25
    x = torch.randn(n, 64, device=dev)
    y = torch.randint(0, 5, (n,), device=dev)
    torch.manual_seed(0)
    m = DiGCN_node_classification(64, 64, 5, 0.5).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=0.01)

    def step():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.nll_loss(m(x, ei, w), y).backward()
        opt.step()
    out["DiGCN_node_classification h=64, 2M / 50M"] = timed(step)
for k, v in out.items():
    print(f"{v:9.2f} ms/step  {k}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/models.json", "w"), indent=1)
