"""Node-range sharding of the path across the GPUs of one node.

The reference has no multi-device story at all, so this is new design, MI355X-first:
one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI); rank g owns the node
range [g*n_pad, (g+1)*n_pad) of every feature matrix and, of each operator, the CSR rows it
PRODUCES: the by-target rows of its nodes for the forward product and the by-source rows of its
nodes for the backward product.  Output rows are independent, so the only data-path exchange is an
all-gather of the [n_pad, 2F] packed (real | imag) feature block before each propagate -- forward:
the layer input, backward: the gradient of the propagated term.  No reduce-scatter, no atomics;
weight gradients are all-reduced (tiny).  Equal-size ranges (the last one zero-padded) keep the
collective a plain `all_gather_into_tensor` that RCCL spreads over all seven xGMI links.

On the benchmark's random SBM graphs the halo is ~every row (a rank's 5M local entries touch 99% of
the 1M source rows), so gathering whole blocks loses nothing against a halo list.

Grid layout (`GridPlan`, the default of `ShardedMagNetConv` when the width allows it).  On a locality-free
graph the row layout is exchange-bound: every propagate moves (P-1)/P of BOTH feature matrices into every GPU
(448 MB at P = 8, ~1.1 ms on xGMI) while the local product shrinks to 0.4 ms.  Gathered feature rows cost
whole 128-byte lines, so a GPU can take a quarter of the COLUMNS of the packed (real | imag) rows at no loss of
gather efficiency (measured, tools/narrow_probe.py: 41 M entries at 16 + 16 packed floats: 0.81 ms = the cost
of one line per entry).  The P ranks therefore form a p_r x p_c grid (p_c <= 4): rank (i, j) multiplies row
block i of the operator (a contiguous slice of the shared CSR, no re-sort) with column slice j of the
features.  Exchange per propagate: an all-to-all that hands every rank the column slice j of all rows (1/p_c of
the all-gather volume), and a second one inside the row group that returns the product to node-range
ownership.  Ownership of inputs, outputs and the dense stage stays node-range, as in the row layout.

`ShardPlan`, `GridPlan`, `all_gather_rows` and `exchange` are device- and backend-agnostic (exercised with gloo on CPU in
tests/test_sharding_gloo.py); `ShardedMagNetConv` is the HIP compute path.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist

Tensor = torch.Tensor


class ShardPlan:
    """Contiguous, equal-size node ranges: rank g owns global ids [g*n_pad, (g+1)*n_pad)."""

    def __init__(self, num_nodes: int, world_size: int, rank: int):
        if not (0 <= rank < world_size):
            raise ValueError(f"rank {rank} outside world of {world_size}")
        self.num_nodes, self.world_size, self.rank = int(num_nodes), int(world_size), int(rank)
        self.n_pad = (self.num_nodes + world_size - 1) // world_size
        self.n_total = self.n_pad * world_size          # padded node count (extra nodes are isolated)
        self.lo = rank * self.n_pad
        self.hi = min(self.lo + self.n_pad, self.num_nodes)
        self.n_local = max(self.hi - self.lo, 0)        # real rows owned by this rank

    def shard_rows(self, x: Tensor) -> Tensor:
        """Rows of a global [num_nodes, F] matrix owned by this rank, zero-padded to n_pad rows."""
        out = x.new_zeros((self.n_pad,) + tuple(x.shape[1:]))
        if self.n_local > 0:
            out[:self.n_local] = x[self.lo:self.hi]
        return out

    def unshard_rows(self, gathered: Tensor) -> Tensor:
        """[n_total, ...] gathered buffer -> the [num_nodes, ...] global matrix."""
        return gathered[:self.num_nodes]

    def owned(self, ids: Tensor) -> Tensor:
        return (ids >= self.lo) & (ids < self.lo + self.n_pad)

    def local_entries(self, edge_index: Tensor, by: int) -> Tuple[Tensor, Tensor]:
        """Entries whose row `by` (0 = source, 1 = target) is owned by this rank.
        Returns (positions in the COO list, the sub-COO with the owned row re-based to local ids)."""
        keep = self.owned(edge_index[by]).nonzero(as_tuple=True)[0]
        sub = edge_index[:, keep].clone()
        sub[by] -= self.lo
        return keep, sub


def all_gather_rows(x_local: Tensor, group=None) -> Tensor:
    """[n_pad, C] per rank -> [world * n_pad, C], rank-major (== global node order under ShardPlan).
    One collective; on RCCL it runs on the calling stream's NCCL stream semantics of torch."""
    world = dist.get_world_size(group)
    x_local = x_local.contiguous()
    out = x_local.new_empty((world * x_local.size(0),) + tuple(x_local.shape[1:]))
    try:
        dist.all_gather_into_tensor(out, x_local, group=group)
    except (RuntimeError, NotImplementedError):  # backends without the fused form
        parts = list(out.chunk(world, dim=0))
        dist.all_gather(parts, x_local, group=group)
    return out


def pack_pair(a: Tensor, b: Tensor) -> Tensor:
    """[n, F], [n, F] -> [n, 2F] (real | imag side by side: one gathered row feeds both operators)."""
    return torch.cat([a, b], dim=1)


class GridPlan(ShardPlan):
    """p_r x p_c process grid over the same node-range ownership as ShardPlan: rank r = i * p_c + j owns node
    block r (n_pad rows); as a worker it multiplies ROW BLOCK i = node blocks [i * p_c, (i + 1) * p_c) of the
    operator with COLUMN SLICE j = features [j * fc, (j + 1) * fc) of both packed operands."""

    def __init__(self, num_nodes: int, world_size: int, rank: int, n_feat: int, p_c: Optional[int] = None):
        super().__init__(num_nodes, world_size, rank)
        self.p_c = self.choose_cols(world_size, n_feat) if p_c is None else int(p_c)
        if world_size % self.p_c or n_feat % self.p_c:
            raise ValueError(f"grid of {self.p_c} column slices does not divide world {world_size} / width {n_feat}")
        self.p_r = world_size // self.p_c
        self.i, self.j = rank // self.p_c, rank % self.p_c
        self.fc = n_feat // self.p_c
        self.n_feat = n_feat
        self.block_rows = self.p_c * self.n_pad                     # rows of operator row block i
        self.block_lo = self.i * self.block_rows

    @staticmethod
    def choose_cols(world_size: int, n_feat: int) -> int:
        """Largest p_c <= 4 dividing the world with 16-byte-aligned column slices (fc % 4 == 0): two packed
        slices of >= 16 floats still fill the 128-byte line a gather costs anyway.  Received volume relative to
        one full feature pair: rows (1 - 1/P); grid (1 - 1/P) / p_c + (1 - 1/p_c) / P -- a 1 x 2 grid on two
        ranks moves exactly what the row layout moves, so two ranks stay in the row layout."""
        for p_c in (4, 2):
            if world_size % p_c == 0 and world_size > p_c - 1 + (p_c == 2) and n_feat % (4 * p_c) == 0:
                return p_c
        return 1

    # -- pure index bookkeeping of the two exchanges (used by the device path and by the CPU tests) ----------
    def slice_chunks(self, a: Tensor, b: Tensor) -> Tensor:
        """Row-layout [n_pad, F] pair -> [P, n_pad, 2 fc]: chunk d = the packed column slice j(d) = d % p_c of
        my rows (the same slice for every rank of one grid column)."""
        n, fc = a.size(0), self.fc
        packed = torch.stack([a.reshape(n, self.p_c, fc), b.reshape(n, self.p_c, fc)], dim=2)   # [n, p_c, 2, fc]
        slices = packed.permute(1, 0, 2, 3).reshape(self.p_c, n, 2 * fc)
        return slices.repeat(self.p_r, 1, 1)

    def group_splits(self):
        """dim-0 split sizes of the exchange inside my row group (1 chunk per member, 0 for everyone else)."""
        g = self.row_group()
        return [1 if d in g else 0 for d in range(self.world_size)]

    def row_group(self):
        return range(self.i * self.p_c, (self.i + 1) * self.p_c)

    def merge_slices(self, recv: Tensor):
        """[p_c, n_pad, 2 fc] (source rank j' of my row group -> its column slice of MY rows) -> row-layout
        pair [n_pad, F], [n_pad, F]."""
        n, fc = recv.size(1), self.fc
        r = recv.reshape(self.p_c, n, 2, fc).permute(2, 1, 0, 3).reshape(2, n, self.p_c * fc)
        return r[0], r[1]


def exchange(out: Tensor, inp: Tensor, out_splits=None, in_splits=None, group=None) -> Tensor:
    """all_to_all_single along dim 0 (splits in dim-0 units, None = equal): chunk d of `inp` goes to rank d,
    chunk s of `out` comes from rank s; a split of 0 = nothing to exchange with that peer.  RCCL runs it as
    one grouped send/recv.  gloo has no device all-to-all, so device tensors are staged through the host there
    (test configurations only)."""
    if dist.get_backend(group) == "gloo" and inp.is_cuda:
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(host)
    else:
        dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=group)
    return out


def collect_slices(plan: "GridPlan", a_loc: Tensor, b_loc: Tensor, group=None) -> Tensor:
    """Row-layout pair [n_pad, F] x 2 -> packed column slice j of ALL rows, [n_total, 2 fc] = (real | imag)."""
    full = a_loc.new_empty((plan.world_size, plan.n_pad, 2 * plan.fc))
    exchange(full, plan.slice_chunks(a_loc, b_loc), group=group)
    return full.view(plan.n_total, 2 * plan.fc)


def return_rows(plan: "GridPlan", ya: Tensor, yb: Tensor, group=None) -> Tuple[Tensor, Tensor]:
    """(row block i, column slice j) products [p_c * n_pad, fc] x 2 -> node-range ownership [n_pad, F] x 2: inside
    the row group, rank (i, j) hands rank (i, j') the rows of node block i * p_c + j' and receives the other
    column slices of its own rows."""
    pack = torch.cat([ya, yb], dim=1).view(plan.p_c, plan.n_pad, 2 * plan.fc)
    recv = pack.new_empty((plan.p_c, plan.n_pad, 2 * plan.fc))
    splits = plan.group_splits()
    exchange(recv, pack, splits, splits, group)
    return plan.merge_slices(recv)


class _ShardedSpmmFn(torch.autograd.Function):
    """y_local = S[my target rows, :] x  with x gathered from all ranks; backward
    dx_local = S^T[my source rows, :] dy with dy gathered.  fp32 or bf16 storage (bf16 halves the
    exchanged bytes as well as the gathered ones)."""

    @staticmethod
    def forward(ctx, x_local, op):
        from .sparse import _spmm_raw
        full = all_gather_rows(x_local, op.group)
        ctx.op = op
        return _spmm_raw(op.fwd_csr, op.fwd_val, full, None, 1.0, 0.0, op.mean)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_local):
        from .sparse import _spmm_raw
        op = ctx.op
        full = all_gather_rows(g_local.contiguous(), op.group)
        return _spmm_raw(op.bwd_csr, op.bwd_val, full, None, 1.0, 0.0, False), None


class ShardedOperator:
    """A COO operator out[scatter] += w * x[gather] sharded by node range: this rank keeps the by-target
    rows of its nodes (forward) and the by-source rows of its nodes (backward), both with GLOBAL column ids
    into the all-gathered feature matrix.  `apply(x_local)` is differentiable w.r.t. x_local."""

    def __init__(self, edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int, group=None,
                 flow: str = "source_to_target", reduce: str = "add"):
        from .sparse import csr_from_coo, gather_values
        self.group = group
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        self.plan = plan = ShardPlan(num_nodes, world, rank)
        g, s = (0, 1) if flow == "source_to_target" else (1, 0)
        coo = torch.stack([edge_index[g], edge_index[s]])            # row 0 = gather (source), row 1 = scatter
        self.mean = reduce == "mean"
        keep_t, sub_t = plan.local_entries(coo, by=1)
        self.fwd_csr = csr_from_coo(sub_t[1], sub_t[0], plan.n_pad, plan.n_total)
        keep_s, sub_s = plan.local_entries(coo, by=0)
        self.bwd_csr = csr_from_coo(sub_s[0], sub_s[1], plan.n_pad, plan.n_total)
        w = edge_weight
        if self.mean:                                                  # backward of mean: 1 / in-degree per entry
            deg = torch.zeros(plan.n_total, dtype=torch.float32, device=coo.device).index_add_(
                0, coo[1], torch.ones(coo.size(1), dtype=torch.float32, device=coo.device)).clamp(min=1)
            inv = (1.0 / deg)[coo[1]]
            wb = inv if w is None else w.float() * inv
        else:
            wb = w
        self.fwd_val = None if w is None else gather_values(w[keep_t], self.fwd_csr.perm)
        self.bwd_val = None if wb is None else gather_values(wb[keep_s], self.bwd_csr.perm)
        self.local_nnz = int(keep_t.numel())

    def apply(self, x_local: Tensor) -> Tensor:
        return _ShardedSpmmFn.apply(x_local, self)


class ShardedDiGCNConv(torch.nn.Module):
    """DiGCNConv (out = S^T (x W) + b, reference nn/directed/DiGCNConv.py:54-94) over a node-range-sharded
    graph; fp32 or bf16 (`.to(torch.bfloat16)`: BASELINE config "DiGCN_Inception_Block ... bf16, 8xMI355X").
    Parameters are replicated; their gradients are all-reduced by hooks during backward."""

    def __init__(self, in_channels: int, out_channels: int, num_nodes: int, edge_index: Tensor,
                 edge_weight: Tensor, bias: bool = True, device=None, group=None):
        super().__init__()
        from .nn import DiGCNConv
        proto = DiGCNConv(in_channels, out_channels, bias=bias)
        self.weight, self.bias = proto.weight, proto.bias
        self.in_channels, self.out_channels = in_channels, out_channels
        device = device or edge_index.device
        self.to(device)
        if edge_weight is None:
            raise RuntimeError('Normalized adj matrix cannot be None. Please obtain the adj matrix in preprocessing.')
        self.op = ShardedOperator(edge_index.to(device), edge_weight.to(device), num_nodes, group)
        self.plan, self.group = self.op.plan, group
        for prm in self.parameters():
            prm.register_hook(self._allreduce)

    def _allreduce(self, grad):
        grad = grad.contiguous()
        dist.all_reduce(grad, group=self.group)
        return grad

    def shard_rows(self, x: Tensor) -> Tensor:
        return self.plan.shard_rows(x)

    def forward(self, x_local: Tensor) -> Tensor:
        from .dense import tall_linear
        out = self.op.apply(tall_linear(x_local, self.weight))
        return out if self.bias is None else out + self.bias


class _ShardedMagneticFn(torch.autograd.Function):
    """Forward / backward of one node-sharded MagNetConv layer (local rows only)."""

    @staticmethod
    def forward(ctx, x_real, x_imag, weight, bias, layer):
        from .dense import dense_fwd_raw
        from .sparse import _spmm2_raw
        k1, f = weight.size(0), x_real.size(1)
        ta, tb = [x_real.contiguous()], [x_imag.contiguous()]
        csr, vr, vi = layer._fwd_csr, layer._fwd_vals[0], layer._fwd_vals[1]
        for k in range(1, k1):
            full = all_gather_rows(pack_pair(ta[k - 1], tb[k - 1]), layer.group)
            if k == 1:
                ya, yb = _spmm2_raw(csr, vr, vi, full[:, :f], full[:, f:], None, None, 1.0, 0.0)
            else:
                ya, yb = _spmm2_raw(csr, vr, vi, full[:, :f], full[:, f:], ta[k - 2], tb[k - 2], 2.0, -1.0)
            ta.append(ya)
            tb.append(yb)
        out_r, out_i = layer._dense_fwd(ta, tb, weight, bias)
        ctx.layer, ctx.k1, ctx.has_bias = layer, k1, bias is not None
        ctx.save_for_backward(weight, *ta, *tb)
        return out_r, out_i

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_r, g_i):
        from .sparse import _spmm2_raw
        layer, k1 = ctx.layer, ctx.k1
        saved = ctx.saved_tensors
        weight = saved[0]
        ta, tb = list(saved[1:1 + k1]), list(saved[1 + k1:1 + 2 * k1])
        da, db, dw, dbias = layer._dense_bwd(ta, tb, weight, g_r, g_i)
        f = da[0].size(1)
        csr, vr, vi = layer._bwd_csr, layer._bwd_vals[0], layer._bwd_vals[1]
        gx_r = gx_i = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            for k in range(k1 - 1, 1, -1):
                full = all_gather_rows(pack_pair(da[k], db[k]), layer.group)
                da[k - 1], db[k - 1] = _spmm2_raw(csr, vr, vi, full[:, :f], full[:, f:], da[k - 1], db[k - 1],
                                                  2.0, 1.0)
                da[k - 2].sub_(da[k])
                db[k - 2].sub_(db[k])
            if k1 > 1:
                full = all_gather_rows(pack_pair(da[1], db[1]), layer.group)
                gx_r, gx_i = _spmm2_raw(csr, vr, vi, full[:, :f], full[:, f:], da[0], db[0], 1.0, 1.0)
            else:
                gx_r, gx_i = da[0], db[0]
        # parameter gradients: sum of the per-shard partials
        dist.all_reduce(dw, group=layer.group)
        if ctx.has_bias:
            dist.all_reduce(dbias, group=layer.group)
        return gx_r, gx_i, dw, (dbias if ctx.has_bias else None), None


class _GridMagneticFn(torch.autograd.Function):
    """One MagNetConv layer in the grid layout (GridPlan): per Chebyshev order one slice exchange, one dual
    SpMM over (row block i) x (column slice j), one exchange back to node-range ownership."""

    @staticmethod
    def forward(ctx, x_real, x_imag, weight, bias, layer):
        k1 = weight.size(0)
        ta, tb = [x_real.contiguous()], [x_imag.contiguous()]
        mine = []                                     # T_k restricted to (row block i, column slice j), packed
        for k in range(1, k1):
            full = layer._collect_slices(ta[k - 1], tb[k - 1])
            mine.append(layer._own_block(full))
            z = mine[k - 2] if k >= 2 else None
            ya, yb = layer._grid_product(full, z, 1.0 if k == 1 else 2.0, 0.0 if k == 1 else -1.0, False)
            ra, rb = layer._return_rows(ya, yb)
            ta.append(ra)
            tb.append(rb)
        out_r, out_i = layer._dense_fwd(ta, tb, weight, bias)
        ctx.layer, ctx.k1, ctx.has_bias = layer, k1, bias is not None
        ctx.save_for_backward(weight, *ta, *tb)
        return out_r, out_i

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_r, g_i):
        layer, k1 = ctx.layer, ctx.k1
        saved = ctx.saved_tensors
        weight = saved[0]
        ta, tb = list(saved[1:1 + k1]), list(saved[1 + k1:1 + 2 * k1])
        da, db, dw, dbias = layer._dense_bwd(ta, tb, weight, g_r, g_i)
        gx_r = gx_i = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            # d T_{k-1} += 2 S^T d T_k ; d T_{k-2} -= d T_k   (k = K .. 2), then gX = d T_0 + S^T d T_1
            for k in range(k1 - 1, 0, -1):
                full = layer._collect_slices(da[k], db[k])
                ya, yb = layer._grid_product(full, None, 2.0 if k >= 2 else 1.0, 0.0, True)
                ra, rb = layer._return_rows(ya, yb)
                da[k - 1] = da[k - 1] + ra
                db[k - 1] = db[k - 1] + rb
                if k >= 2:
                    da[k - 2] = da[k - 2] - da[k]
                    db[k - 2] = db[k - 2] - db[k]
            gx_r, gx_i = da[0], db[0]
        dist.all_reduce(dw, group=layer.group)
        if ctx.has_bias:
            dist.all_reduce(dbias, group=layer.group)
        return gx_r, gx_i, dw, (dbias if ctx.has_bias else None), None


class ShardedMagNetConv(torch.nn.Module):
    """MagNetConv over a node-range-sharded graph: each rank owns the rows [lo, hi) of the features (inputs,
    outputs, dense stage).  Parameters are replicated (same seed => same init on every rank); their gradients
    come back all-reduced.  forward(x_real_local, x_imag_local) -> local output rows.

    layout = "rows": every rank multiplies its own operator rows with the all-gathered features.
    layout = "grid": p_r x p_c process grid (GridPlan, see the module docstring) -- 1 / p_c of the exchange
                     volume at the same gather efficiency.
    layout = "auto" (default): grid whenever the input width splits into 16-byte-aligned column slices.
    """

    def __init__(self, in_channels: int, out_channels: int, K: int, q: float, num_nodes: int,
                 edge_index: Tensor, edge_weight: Optional[Tensor] = None, normalization: str = "sym",
                 bias: bool = True, device=None, group=None, signed: bool = False,
                 absolute_degree: bool = True, layout: str = "auto", grid_cols: Optional[int] = None):
        super().__init__()
        from .nn import MagNetConv, MSConv
        from .sparse import csr_from_coo, gather_values
        self.group = group
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if layout not in ("auto", "rows", "grid"):
            raise ValueError(f"unknown layout {layout!r}")
        p_c = grid_cols if grid_cols is not None else GridPlan.choose_cols(world, in_channels)
        if layout == "auto":
            layout = "grid" if (p_c > 1 and in_channels % (4 * p_c) == 0) else "rows"
        self.layout = layout
        self.plan = GridPlan(num_nodes, world, rank, in_channels, p_c) if layout == "grid" \
            else ShardPlan(num_nodes, world, rank)
        device = device or edge_index.device
        proto = (MSConv(in_channels, out_channels, K, q, False, normalization, bias, True, absolute_degree)
                 if signed else MagNetConv(in_channels, out_channels, K, q, False, normalization, True, bias))
        self.weight = proto.weight
        self.bias = proto.bias
        self.in_channels, self.out_channels = in_channels, out_channels
        self.to(device)
        # every rank builds the global operator (degrees are global), padded to n_total isolated-extended
        # nodes, then keeps only the entries whose produced row it owns
        lam = torch.tensor(2.0, dtype=torch.float32, device=device)
        op = proto._build_operator(edge_index.to(device), self.plan.n_total,
                                   None if edge_weight is None else edge_weight.to(device), q, normalization,
                                   lam, torch.float32)
        if layout == "grid":
            self._init_grid(op, num_nodes)
            return
        coo, vr, vi = op.coo()                                            # row 0 = source, row 1 = target
        self.global_nnz = int(coo.size(1)) - (self.plan.n_total - num_nodes)
        n_tot, n_pad = self.plan.n_total, self.plan.n_pad
        keep_t, sub_t = self.plan.local_entries(coo, by=1)   # forward: rows = my targets, cols = sources
        self._fwd_csr = csr_from_coo(sub_t[1], sub_t[0], n_pad, n_tot)
        keep_s, sub_s = self.plan.local_entries(coo, by=0)   # backward: rows = my sources, cols = targets
        self._bwd_csr = csr_from_coo(sub_s[0], sub_s[1], n_pad, n_tot)
        self._fwd_vals = (gather_values(vr[keep_t], self._fwd_csr.perm), gather_values(vi[keep_t], self._fwd_csr.perm))
        self._bwd_vals = (gather_values(vr[keep_s], self._bwd_csr.perm), gather_values(vi[keep_s], self._bwd_csr.perm))
        self.local_nnz = int(keep_t.numel())
        del op

    # ---- grid layout -------------------------------------------------------------------------------
    def _init_grid(self, op, num_nodes):
        """Row block i of the shared CSR (the operator's pattern is symmetric, so the same slice with the
        mirrored values is row block i of the transposed operator): a contiguous range of rowptr / col /
        values -- no re-sort."""
        from .sparse import CSR
        plan = self.plan
        csr = op.csr
        lo, hi = plan.block_lo, plan.block_lo + plan.block_rows
        e0, e1 = int(csr.rowptr[lo]), int(csr.rowptr[hi])
        rowptr = (csr.rowptr[lo:hi + 1] - e0).contiguous()
        self._grid_csr = CSR(plan.block_rows, plan.n_total, e1 - e0, rowptr, csr.col[e0:e1].contiguous(), None)
        self._grid_fwd_vals = tuple(v[e0:e1].contiguous() for v in op.values_fwd)
        self._grid_bwd_vals = tuple(v[e0:e1].contiguous() for v in op.values_bwd)
        self.global_nnz = int(csr.nnz) - (plan.n_total - num_nodes)
        self.local_nnz = e1 - e0

    def _collect_slices(self, a_loc: Tensor, b_loc: Tensor) -> Tensor:
        return collect_slices(self.plan, a_loc, b_loc, self.group)

    def _own_block(self, full: Tensor) -> Tensor:
        plan = self.plan
        return full[plan.block_lo:plan.block_lo + plan.block_rows]

    def _grid_product(self, full: Tensor, z: Optional[Tensor], alpha: float, beta: float, transposed: bool):
        from .sparse import _spmm2_raw
        fc = self.plan.fc
        vr, vi = self._grid_bwd_vals if transposed else self._grid_fwd_vals
        za, zb = (None, None) if z is None else (z[:, :fc], z[:, fc:])
        return _spmm2_raw(self._grid_csr, vr, vi, full[:, :fc], full[:, fc:], za, zb, alpha, beta)

    def _return_rows(self, ya: Tensor, yb: Tensor):
        return return_rows(self.plan, ya, yb, self.group)

    # dense stage: the fused MFMA kernels when the shape is tiled by them, library GEMMs otherwise
    def _dense_fwd(self, ta, tb, weight, bias):
        from .dense import dense_fwd_raw, dense_supported
        if dense_supported(self.in_channels, self.out_channels, weight.size(0)):
            return dense_fwd_raw(ta, tb, weight, bias)
        rr = sum(torch.matmul(ta[k], weight[k]) for k in range(weight.size(0)))
        ii = sum(torch.matmul(tb[k], weight[k]) for k in range(weight.size(0)))
        b = 0 if bias is None else bias
        return rr - ii + b, rr + ii + b

    def _dense_bwd(self, ta, tb, weight, g_r, g_i):
        from .dense import dense_bwd_raw, dense_supported
        if dense_supported(self.in_channels, self.out_channels, weight.size(0)):
            return dense_bwd_raw(ta, tb, weight, g_r, g_i)
        p, m = g_r + g_i, g_i - g_r
        k1 = weight.size(0)
        da = [torch.matmul(p, weight[k].t()) for k in range(k1)]
        db = [torch.matmul(m, weight[k].t()) for k in range(k1)]
        dw = torch.stack([ta[k].t() @ p + tb[k].t() @ m for k in range(k1)])
        return da, db, dw, p.sum(0)

    def shard_rows(self, x: Tensor) -> Tensor:
        return self.plan.shard_rows(x)

    def forward(self, x_real_local: Tensor, x_imag_local: Tensor):
        fn = _GridMagneticFn if self.layout == "grid" else _ShardedMagneticFn
        return fn.apply(x_real_local, x_imag_local, self.weight, self.bias, self)

    def allreduce_grads(self):
        """Kept for API symmetry: parameter gradients are already all-reduced inside backward."""
        return None
