"""Probe (GPU box): what a write-only stream gets on this part -- torch fill_ / zero_ (hipMemset path), and a 3:1 write:read mix
(out[3 n] = broadcast of in[n], the shape of tall_linear_bf16<2,12>) -- next to the streaming copy.  GB/s of bytes moved."""
import torch, time, json, os
dev = torch.device("cuda:0")
n = 512 * 1024 * 1024          # floats: 2 GiB
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)
def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters
out = {}
ms = timed(lambda: x.fill_(1.5)); out["fill_2GiB_GBps"] = 4 * n / ms / 1e6
ms = timed(lambda: x.zero_()); out["zero_2GiB_GBps"] = 4 * n / ms / 1e6
ms = timed(lambda: y.copy_(x)); out["copy_2GiB_GBps_read_plus_write"] = 8 * n / ms / 1e6
q = n // 4
src = x[:q].view(-1, 64)
dst = y[:3 * q].view(-1, 3, 64)
ms = timed(lambda: dst.copy_(src.unsqueeze(1).expand(-1, 3, -1))); out["write3_read1_GBps"] = 16 * q / ms / 1e6
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/write_probe.json", "w"), indent=1)
