"""Measurement helper: fused MFMA dense stage (forward / backward) at the benchmark widths."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_signed_directed_amd.dense import dense_bwd_raw, dense_fwd_raw
dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


n = 1000000
for f, k1 in ((64, 2), (128, 3), (128, 2)):
    ta = [torch.randn(n, f, device=dev) for _ in range(k1)]
    tb = [torch.randn(n, f, device=dev) for _ in range(k1)]
    w = torch.randn(k1, f, f, device=dev) * 0.1
    bias = torch.randn(f, device=dev)
    gr, gi = torch.randn(n, f, device=dev), torch.randn(n, f, device=dev)
    tf = timeit(lambda: dense_fwd_raw(ta, tb, w, bias))
    tb_ = timeit(lambda: dense_bwd_raw(ta, tb, w, gr, gi))
    gflop = n * k1 * f * f * 4 / 1e9
    print(f"N={n} F={f} K+1={k1}: fwd {tf:.3f} ms ({gflop / tf:.0f} TFLOP/s... {gflop:.0f} GFLOP), "
          f"bwd {tb_:.3f} ms ({2 * gflop / tb_:.0f} GFLOP/ms)", flush=True)
