"""Where does the rows-per-wavefront SpMM variant (spmm_packed_kernel: every LPR-lane group owns its own row) beat
one wavefront per row?  Sweeps entries per row x width, single and dual operator, on ~20 M entries with Poisson row
lengths (random targets) and 1 M source rows; PYGSD_SPMM_PACKED=0 / 1 forces the variant.  Prints ms per launch and
writes gpurun_out/packed_probe.json; launch_spmm's packed_threshold() is set from this table."""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_geometric_signed_directed_amd.sparse import Pattern, _spmm2_raw, _spmm_raw  # noqa: E402


def timed(fn, reps=7):
    ts = []
    for k in range(reps + 2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        if k >= 2:
            ts.append(a.elapsed_time(b))
    return statistics.median(ts)


def main():
    dev = torch.device("cuda:0")
    n_cols, nnz = 1000000, 20000000
    out = {}
    g = torch.Generator(device="cuda").manual_seed(0)
    for deg in (4, 6, 8, 12, 16, 24, 32, 48, 64):
        n_rows = nnz // deg
        ei = torch.stack([torch.randint(0, n_cols, (nnz,), device=dev, generator=g),
                          torch.randint(0, n_rows, (nnz,), device=dev, generator=g)])
        csr = Pattern(ei, n_cols, n_rows).fwd
        va = torch.rand(nnz, device=dev, generator=g)
        vb = torch.rand(nnz, device=dev, generator=g)
        for f in (16, 32, 64):
            xa = torch.randn(n_cols, f, device=dev, generator=g)
            xb = torch.randn(n_cols, f, device=dev, generator=g)
            for name, fn in (("single", lambda: _spmm_raw(csr, va, xa, None, 1.0, 0.0, False)),
                             ("dual", lambda: _spmm2_raw(csr, va, vb, xa, xb, None, None, 1.0, 0.0))):
                rec = {}
                for mode, v1, key in (("0", "0", "row_per_wave_ms"), ("1", "0", "packed_ms"), ("1", "1", "packed_v1_ms")):
                    os.environ["PYGSD_SPMM_PACKED"] = mode
                    os.environ["PYGSD_SPMM_PACKED_V1"] = v1           # round 3's kernel (per-group col / val loads)
                    rec[key] = timed(fn)
                os.environ.pop("PYGSD_SPMM_PACKED")
                os.environ.pop("PYGSD_SPMM_PACKED_V1")
                rec["packed_speedup"] = rec["row_per_wave_ms"] / rec["packed_ms"]
                rec["packed_vs_v1"] = rec["packed_v1_ms"] / rec["packed_ms"]
                out[f"deg{deg}_F{f}_{name}"] = rec
                print(f"deg {deg:3d} F {f:3d} {name:6s}: row/wave {rec['row_per_wave_ms']:.3f} ms  packed {rec['packed_ms']:.3f} ms  "
                      f"(round 3's: {rec['packed_v1_ms']:.3f})  x{rec['packed_speedup']:.2f}", flush=True)
            del xa, xb
        del csr, ei, va, vb
        torch.cuda.empty_cache()
    with open(os.path.join(ROOT, "gpurun_out", "packed_probe.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
