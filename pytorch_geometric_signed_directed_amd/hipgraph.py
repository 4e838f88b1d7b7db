"""hipGraph capture of a whole train / inference step.

On small graphs (the reference's Cora-ML-sized examples: ~3 k nodes, ~8 k edges) a step is a few dozen
kernels of microseconds each and the host launch path dominates (measured: 1.19 ms eager vs 0.45 ms replayed,
profiles/r1_small_graph_hipgraph.json).  The C-ABI kernels are plain launches on the caller's stream, so
`hipStreamBeginCapture` (driven through torch.cuda.graphs) records them like any other kernel and one
`hipGraphLaunch` replays the whole step.

    step = capture_step(fn)         # fn(): zero_grad / forward / loss.backward() / optimizer.step()
    for _ in range(epochs):
        loss = step()               # replays; returns fn's outputs (static tensors, updated in place)

Requirements (those of any captured region): fixed shapes and tensors -- inputs are updated by copying into
the tensors `fn` closes over; layers constructed with `cached=True` (or called with unchanged graph tensors)
so that no operator is rebuilt inside the captured region (a rebuild reads sizes back to the host);
optimisers created with `capturable=True`; no `.item()` / host reads inside `fn`.
"""
from typing import Any, Callable

import torch


class CapturedStep:
    def __init__(self, fn: Callable[[], Any], warmup: int = 3):
        if not torch.cuda.is_available():
            raise RuntimeError("capture_step needs a GPU (hipGraph capture)")
        self.fn = fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):           # warm-up off the capture: builds cached operators, grows the pools
            for _ in range(max(warmup, 1)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn()

    def __call__(self):
        self.graph.replay()
        return self.outputs


def capture_step(fn: Callable[[], Any], warmup: int = 3) -> CapturedStep:
    """Run `fn` `warmup` times eagerly on a side stream, capture one more call into a hipGraph, and return a
    callable that replays it.  Note the warm-up and the capture pass DO execute `fn` (optimizer steps included)."""
    return CapturedStep(fn, warmup)
