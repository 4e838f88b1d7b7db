"""Test infrastructure for the CPU (gloo) runs of the sharded path: torch restatements of the two product
kernels (same signatures as parallel._hip_dual / _hip_single) and of the device-side CSR builders, and the
operator rows of a sharded magnetic layer taken from the ORACLE's operator (the HIP build cannot run without a
GPU).  Everything else -- plan, phases, row chunks, packing, exchanges, merges, the autograd wrappers -- is the
production code of pytorch_geometric_signed_directed_amd/parallel.py."""
import torch

from oracle import ref_layers as R
from pytorch_geometric_signed_directed_amd.parallel import take_rows
from pytorch_geometric_signed_directed_amd.sparse import CSR


def cpu_single(csr, val, x, y, lo, hi, alpha, accumulate, mean):
    rp = csr.rowptr.long()
    e0, e1 = int(rp[lo]), int(rp[hi])
    counts = rp[lo + 1:hi + 1] - rp[lo:hi]
    rows = torch.repeat_interleave(torch.arange(hi - lo), counts)
    v = torch.ones(e1 - e0) if val is None else val[e0:e1].float()
    contrib = x[csr.col[e0:e1].long()].float() * v[:, None]
    out = torch.zeros(hi - lo, x.size(1)).index_add_(0, rows, contrib)
    if mean:
        out = out / counts.clamp(min=1)[:, None]
    out = alpha * out
    if accumulate:
        out = out + y[lo:hi].float()
    y[lo:hi] = out.to(y.dtype)


def cpu_dual(csr, va, vb, xa, xb, ya, yb, lo, hi, alpha, accumulate):
    cpu_single(csr, va, xa, ya, lo, hi, alpha, accumulate, False)
    cpu_single(csr, vb, xb, yb, lo, hi, alpha, accumulate, False)


KERNELS = (cpu_dual, cpu_single)


def cpu_csr_from_coo(seg, other, n_seg, n_other, validate=True):
    """Stable grouping by `seg` -- what pygsd_csr_from_coo does on the device."""
    order = torch.sort(seg, stable=True).indices
    rowptr = torch.zeros(n_seg + 1, dtype=torch.int32)
    rowptr[1:] = torch.bincount(seg, minlength=n_seg).cumsum(0).to(torch.int32)
    return CSR(n_seg, n_other, int(seg.numel()), rowptr, other[order].to(torch.int32), order.to(torch.int32))


def cpu_gather_values(src, perm):
    return src[perm.long()].float()


def patch_device_builders():
    """ShardedOperator builds its CSRs with the HIP kernels; route them to the restatements above."""
    import pytorch_geometric_signed_directed_amd.sparse as S
    S.csr_from_coo = cpu_csr_from_coo
    S.gather_values = cpu_gather_values


def _csr_of(rows, cols, vals, n):
    """Coalesced (sum of duplicates) CSR of the entries (rows, cols) with [nnz, k] values."""
    m = torch.sparse_coo_tensor(torch.stack([rows, cols]), vals, (n, n) + tuple(vals.shape[1:])).coalesce()
    idx, v = m.indices(), m.values()
    rowptr = torch.zeros(n + 1, dtype=torch.int32)
    rowptr[1:] = torch.bincount(idx[0], minlength=n).cumsum(0).to(torch.int32)
    return CSR(n, n, int(idx.size(1)), rowptr, idx[1].to(torch.int32), None), v


def oracle_operator_rows(edge_index, edge_weight, n, q, signed=False, absolute_degree=True, normalization="sym",
                         lambda_max=2.0):
    """-> callback(plan, engine) for ShardedMagNetConv(operator_rows=...): the rows `engine.block_row_ids` of the
    oracle's scaled magnetic operator in padded ids, real and imaginary parts on one pattern, by target (forward)
    and by source (backward)."""
    ei_r, ei_i, w_r, w_i = R.magnet_operator(edge_index, edge_weight, n, q, normalization, lambda_max, signed=signed,
                                             absolute_degree=absolute_degree)

    def build(plan, engine):
        src = torch.cat([plan.to_padded(ei_r[0]), plan.to_padded(ei_i[0])])
        tgt = torch.cat([plan.to_padded(ei_r[1]), plan.to_padded(ei_i[1])])
        zr, zi = torch.zeros_like(w_r), torch.zeros_like(w_i)
        vals = torch.cat([torch.stack([w_r, zr], 1), torch.stack([zi, w_i], 1)])
        fwd, vf = _csr_of(tgt, src, vals, plan.n_total)          # out[target] += w * x[source]
        bwd, vb = _csr_of(src, tgt, vals, plan.n_total)          # dx[source] += w * dy[target]
        assert torch.equal(fwd.rowptr, bwd.rowptr) and torch.equal(fwd.col, bwd.col)    # symmetric pattern
        rows = engine.block_row_ids("cpu")
        sub, picked = take_rows(fwd, (vf[:, 0].contiguous(), vf[:, 1].contiguous(), vb[:, 0].contiguous(),
                                      vb[:, 1].contiguous()), rows)
        return sub, (picked[0], picked[1]), (picked[2], picked[3]), fwd.nnz

    return build


# ------------------------------------------------------------------------------------------------
# ranks as threads of one process (parallel.ThreadExchange)
# ------------------------------------------------------------------------------------------------
class FunctionCtx:
    """Stand-in for the autograd context of a torch.autograd.Function, to drive forward / backward by hand: the
    autograd engine runs every CUDA backward on one worker thread per device, so collectives inside a backward cannot
    rendezvous when the ranks are threads."""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = tuple(needs_input_grad)
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


def sharded_magnetic_step(layer, x_real, x_imag, g_real, g_imag):
    """Forward + backward of one ShardedMagNetConv on local rows without the autograd engine:
    -> (out_real, out_imag, dx_real, dx_imag, dW, db)."""
    from pytorch_geometric_signed_directed_amd.parallel import _ShardedMagneticFn as Fn
    ctx = FunctionCtx((True, True, True, layer.bias is not None, False))
    with torch.no_grad():
        o_r, o_i = Fn.forward(ctx, x_real, x_imag, layer.weight, layer.bias, layer)
        gx_r, gx_i, dw, db, _ = Fn.backward(ctx, g_real, g_imag)
    return o_r, o_i, gx_r, gx_i, dw, db


def run_ranks_as_threads(world, body):
    """body(rank, exchange) in `world` threads sharing ThreadExchange.create(world); -> [result of rank r].  The first
    exception of any rank is re-raised here (the others are released from their rendezvous)."""
    import threading
    from pytorch_geometric_signed_directed_amd.parallel import ThreadExchange
    exchanges = ThreadExchange.create(world)
    results, errors = [None] * world, []

    def runner(rank):
        try:
            if torch.cuda.is_available():
                torch.cuda.set_device(0)
            results[rank] = body(rank, exchanges[rank])
        except threading.BrokenBarrierError:
            pass
        except BaseException as exc:  # noqa: BLE001
            errors.append((rank, exc))
            exchanges[rank]._sh.barrier.abort()
    threads = [threading.Thread(target=runner, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0][1]
    return results
