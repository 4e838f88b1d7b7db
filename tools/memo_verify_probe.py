import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from pytorch_geometric_signed_directed_amd import memo, graphs, _cabi
from pytorch_geometric_signed_directed_amd.nn import SIMPA
dev = torch.device("cuda:0")
n, entries, h = 500000, 10000000, 64
p = (entries / 2) / (n * (n - 1) / 2)
ei_np, sign, _ = graphs.ssbm(n, 5, p, 0.1, 2.0, seed=2)
ei, sign = torch.from_numpy(ei_np).to(dev), torch.from_numpy(sign).to(dev)
pos, neg = ei[:, sign > 0].contiguous(), ei[:, sign < 0].contiguous()
g = torch.Generator().manual_seed(0)
simpa = SIMPA(2, 0.5).to(dev)
wp = torch.ones(pos.size(1), device=dev); wn = torch.ones(neg.size(1), device=dev)
xp = torch.randn(n, h, generator=g).to(dev).requires_grad_(); xn = torch.randn(n, h, generator=g).to(dev).requires_grad_()
calls = [0]
orig = memo.check_unchanged
def counted(*t):
    calls[0] += 1
    return orig(*t)
memo.check_unchanged = counted
def step():
    simpa.zero_grad(set_to_none=True); xp.grad = xn.grad = None
    simpa(pos, wp, neg, wn, xp, xn).sum().backward()
for verify in (True, False, True):
    memo.set_verify(verify)
    for _ in range(3): step()
    torch.cuda.synchronize(); calls[0] = 0
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    print("verify", verify, "ms/step %.3f" % ((time.perf_counter() - t0) / 10 * 1e3), "check_unchanged calls/step", calls[0] / 10)
