"""Measurement helper (GPU box): the magnetic dense stage's backward in its two arithmetic forms (split: three bf16 pieces per
operand on the bf16 matrix pipe, the default where the shape has it; exact: fmaf chains on the fp32 MFMA) -- time at the
north-star shape with a dense and with a broadcast upstream gradient, and the error of both forms against float64 on a sample of
rows, relative to the natural scale of each output (sum of |terms|)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_geometric_signed_directed_amd.dense import dense_bwd_raw, dense_fwd_raw, set_dense_f32_exact
dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def errors(res, ta, tb, w, gr, gi, rows):
    """max |got - float64| / scale over da, db (sampled rows) and dW, dbias (all rows), scale = the same sums with |.| inside."""
    da, db, dw, dbias = res
    p64, m64 = (gr + gi).double(), (gi - gr).double()
    w64 = w.double()
    out = {}
    worst = 0.0
    for k in range(w.size(0)):
        wa = w64[k].t()
        for got, src in ((da[k], p64), (db[k], m64)):
            want = src[rows] @ wa
            scale = src[rows].abs() @ wa.abs()
            worst = max(worst, float(((got[rows].double() - want).abs() / scale).max()))
    out["dA_dB"] = worst
    worst = 0.0
    for k in range(w.size(0)):
        a64, b64 = ta[k].double(), tb[k].double()
        want = a64.t() @ p64 + b64.t() @ m64
        scale = a64.abs().t() @ p64.abs() + b64.abs().t() @ m64.abs()
        worst = max(worst, float(((dw[k].double() - want).abs() / scale).max()))
    out["dW"] = worst
    out["dbias"] = float(((dbias.double() - p64.sum(0)).abs() / p64.abs().sum(0)).max())
    return out


def fwd_errors(res, ta, tb, w, bias, rows):
    o_r, o_i = res
    want_r = sum((ta[k][rows].double() - tb[k][rows].double()) @ w[k].double() for k in range(w.size(0))) + bias.double()
    want_i = sum((ta[k][rows].double() + tb[k][rows].double()) @ w[k].double() for k in range(w.size(0))) + bias.double()
    scale = sum((ta[k][rows].double().abs() + tb[k][rows].double().abs()) @ w[k].double().abs() for k in range(w.size(0))) \
        + bias.double().abs()
    return max(float(((o_r[rows].double() - want_r).abs() / scale).max()), float(((o_i[rows].double() - want_i).abs() / scale).max()))


out = {"cases": [], "forward": []}
for n, f, k1 in ((1000000, 64, 2), (1000000, 128, 3), (1000003, 128, 2), (37, 64, 3)):
    torch.manual_seed(n + f)
    ta = [torch.randn(n, f, device=dev) for _ in range(k1)]
    tb = [torch.randn(n, f, device=dev) for _ in range(k1)]
    w = torch.randn(k1, f, f, device=dev) / f ** 0.5
    bias = torch.randn(f, device=dev)
    rows = torch.randint(0, n, (min(n, 4096),), device=dev)
    row = {"n": n, "f": f, "k1": k1}
    for name, exact in (("split", False), ("exact", True)):
        prev = set_dense_f32_exact(exact)
        try:
            res = dense_fwd_raw(ta, tb, w, bias)
            row[name] = {"error_vs_float64": fwd_errors(res, ta, tb, w, bias, rows)}
            if n >= 1000000:
                row[name]["ms"] = round(timeit(lambda: dense_fwd_raw(ta, tb, w, bias)), 4)
        finally:
            set_dense_f32_exact(prev)
    out["forward"].append(row)
    del ta, tb
torch.cuda.empty_cache()
for n, f, k1, timed in ((1000000, 64, 2, True), (1000000, 128, 3, True), (100003, 64, 2, False), (37, 128, 2, False)):
    torch.manual_seed(n)
    ta = [torch.randn(n, f, device=dev) for _ in range(k1)]
    tb = [torch.randn(n, f, device=dev) for _ in range(k1)]
    w = torch.randn(k1, f, f, device=dev) * 0.1
    gr, gi = torch.randn(n, f, device=dev), torch.randn(n, f, device=dev)
    one = torch.ones(1, f, device=dev)
    rows = torch.randint(0, n, (min(n, 4096),), device=dev)
    row = {"n": n, "f": f, "k1": k1}
    for name, exact in (("split", False), ("exact", True)):
        prev = set_dense_f32_exact(exact)
        try:
            res = dense_bwd_raw(ta, tb, w, gr, gi)
            row[name] = {"error_vs_float64": errors(res, ta, tb, w, gr, gi, rows)}
            if timed:
                row[name]["dense_gradient_ms"] = round(timeit(lambda: dense_bwd_raw(ta, tb, w, gr, gi)), 4)
                row[name]["broadcast_gradient_ms"] = round(timeit(
                    lambda: dense_bwd_raw(ta, tb, w, one.expand(n, f), one.expand(n, f))), 4)
        finally:
            set_dense_f32_exact(prev)
    out["cases"].append(row)
print(json.dumps(out, indent=1))
