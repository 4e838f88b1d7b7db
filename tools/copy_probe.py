"""Which streaming-copy shape reaches the achievable HBM rate on this part (MI355X_MICROARCH.md quotes 6.29 TB/s
for a float4 copy)?  Sweeps PYGSD_COPY_MODE x PYGSD_COPY_BLOCKS_PER_CU of pygsd_stream_copy_f32 and torch's own
copy_ on 1 GiB; prints GB/s (read + write)."""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_geometric_signed_directed_amd import _cabi  # noqa: E402


def timed(fn, reps=9):
    ts = []
    for k in range(reps + 2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        if k >= 2:
            ts.append(a.elapsed_time(b))
    return statistics.median(ts)


def main():
    n = (1 << 30) // 4
    src = torch.randn(n, device="cuda")
    dst = torch.empty_like(src)
    lib = _cabi.lib()
    out = {"torch_copy_": 2 * n * 4 / (timed(lambda: dst.copy_(src)) * 1e-3) / 1e9}
    for mode in (0, 1, 2, 3):
        for per_cu in (4, 8, 16, 32, 64, 4096):
            os.environ["PYGSD_COPY_MODE"], os.environ["PYGSD_COPY_BLOCKS_PER_CU"] = str(mode), str(per_cu)
            ms = timed(lambda: _cabi.check(lib.pygsd_stream_copy_f32(_cabi.ptr(src), _cabi.ptr(dst), n, _cabi.stream_ptr()), "copy"))
            out[f"mode{mode}_blocks_per_cu{per_cu}"] = 2 * n * 4 / (ms * 1e-3) / 1e9
    print(json.dumps(out, indent=1))
    with open(os.path.join(ROOT, "gpurun_out", "copy_probe.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
