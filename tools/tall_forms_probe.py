"""Measurement helper (GPU box): pygsd_tall_linear's two fp32 forms (split: three bf16 pieces per value on the bf16 matrix pipe, the
default; exact: an fmaf chain per output) at the C3a / C5a shapes -- time per product and error against float64 relative to
sum |x| |w| per output, on rows sampled from the operand."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_signed_directed_amd.dense import set_tall_f32_exact, tall_product
dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


out = []
for n, k, f in ((500000, 128, 64), (500000, 64, 128), (2000000, 64, 192), (2000000, 192, 64), (4099, 256, 64)):
    torch.manual_seed(k + f)
    x = torch.randn(n, k, device=dev)
    w = torch.randn(k, f, device=dev) / k ** 0.5
    rows = torch.randint(0, n, (4096,), device=dev)
    want = x[rows].double() @ w.double()
    scale = x[rows].double().abs() @ w.double().abs()
    row = {"n": n, "K": k, "f_out": f}
    for name, exact in (("split", False), ("exact", True)):
        prev = set_tall_f32_exact(exact)
        try:
            y = tall_product([x], w, False, None)
            row[name] = {"error_vs_float64": float(((y[rows].double() - want).abs() / scale).max()),
                         "ms": round(timeit(lambda: tall_product([x], w, False, None)), 4)}
        finally:
            set_tall_f32_exact(prev)
    out.append(row)
    del x
print(json.dumps(out, indent=1))
