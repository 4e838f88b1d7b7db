"""Measurement helper (GPU box): BASELINE configs[0]-sized model (MagNet_node_classification, 2 layers,
h=16, K=1 on a Cora-ML-sized DSBM: 2995 nodes / 8416 edges / 2879 raw features / 7 classes).  At this size
a train step is launch-bound, so the whole step (forward, loss, backward, Adam) is also captured in a
hipGraph (torch.cuda.graphs drives hipStreamBeginCapture; the C-ABI kernels are plain stream launches and
are captured like any other) and replayed.  Writes gpurun_out/small_graph.json."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
from pytorch_geometric_signed_directed_amd import graphs  # noqa: E402
from pytorch_geometric_signed_directed_amd.nn import MagNet_node_classification  # noqa: E402

dev = torch.device("cuda:0")
n, e, f_in, h, c = 2995, 8416, 2879, 16, 7
ei = torch.from_numpy(graphs.dsbm_for_edges(n, e, k=5, seed=0)[0]).to(dev)
g = torch.Generator().manual_seed(0)
x = torch.randn(n, f_in, generator=g).to(dev)
y = torch.randint(0, c, (n,), generator=g).to(dev)
torch.manual_seed(0)
model = MagNet_node_classification(f_in, hidden=h, K=1, label_dim=c, layer=2, dropout=0.0, cached=True).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)


def step():
    opt.zero_grad(set_to_none=False)
    loss = torch.nn.functional.nll_loss(model(x, x, ei), y)
    loss.backward()
    opt.step()
    return loss


for _ in range(5):
    step()          # builds + caches the operators, warms the allocator
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
eager_ms = (time.perf_counter() - t0) / 50 * 1e3

from pytorch_geometric_signed_directed_amd.hipgraph import capture_step  # noqa: E402
replay = capture_step(step, warmup=3)
graph, static_loss = replay.graph, replay.outputs
torch.cuda.synchronize()
ref = [p.detach().clone() for p in model.parameters()]
graph.replay()
torch.cuda.synchronize()
changed = any(not torch.equal(a, b.detach()) for a, b in zip(ref, model.parameters()))
t0 = time.perf_counter()
for _ in range(200):
    graph.replay()
torch.cuda.synchronize()
graph_ms = (time.perf_counter() - t0) / 200 * 1e3
out = {"config": f"MagNet_node_classification 2 layers h={h} K=1, DSBM {n} nodes / {ei.size(1)} edges, F_in={f_in}, {c} classes",
       "eager_ms_per_train_step": eager_ms, "hipgraph_ms_per_train_step": graph_ms,
       "graph_replay_updates_parameters": bool(changed), "loss_after": float(static_loss)}
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/small_graph.json", "w"), indent=1)
