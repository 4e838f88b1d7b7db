"""Node-range sharding of the path across the GPUs of one node.

The reference has no multi-device story at all (no torch.distributed anywhere in it), so this is new design,
MI355X-first: one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI).

Ownership.  Rank g OWNS a contiguous node range [bounds[g], bounds[g+1]) of every feature matrix (layer inputs,
outputs, the dense stage and its dW / db partials are row-local).  The ranges are chosen for equal WORK, not
equal size (`balanced_bounds`: prefix sums of 1 + in-degree + out-degree), then padded to one common length
n_pad so that every collective is an equal-split one; node v of range g lives at PADDED id g * n_pad + (v -
bounds[g]) and the pad rows are isolated nodes whose features stay zero.

Work layouts for the propagate  Y = S X  (and dX = S^T dY, which for the Hermitian / symmetric operators of
this package is the same pattern with mirrored values):

  rows   rank g multiplies ITS OWN operator rows with the all-gathered packed features -- one all-gather in,
         nothing back.  The right shape for 2 ranks and for one-operand bf16 operators (a 128-byte row is
         already one cache line, column slices would only add lines).
  grid   the P ranks form a p_r x p_c grid; rank (i, j) multiplies ROW BLOCK i -- the i-th 1/p_r of EVERY rank's
         range, so the products go back over all P-1 links, not only to p_c-1 neighbours -- with COLUMN SLICE j of
         the packed (real | imag) features.  One all-to-all in (1 / p_c of the all-gather volume) and one
         all-to-all back.  A gathered feature row costs whole 128-byte lines, so 16 + 16 packed floats cost
         what 64 + 64 cost per line: the grid divides the exchange by p_c at the row layout's compute per rank.

Overlap (both layouts).  xGMI is point-to-point: a collective finishes when the slowest LINK has moved its
block, all links in parallel, so blocks do not arrive one peer after the other and splitting by peer buys
nothing.  The propagate is therefore pipelined along the node dimension instead:
  * every rank's range is cut into C PHASES; the operator's columns are split the same way (`split_phases`),
    phase c's exchange moves sub-range c of every rank, and the partial product over column block c runs while
    phase c + 1 is on the wire (accumulated through the SpMM's own beta * Z epilogue, Z = Y, no atomics);
  * in the grid the last phase's product runs in R ROW CHUNKS and chunk r travels back while chunk r + 1 is
    multiplied.
All exchanges are issued up front as asynchronous collectives (RCCL runs them in order on the process
group's own stream); the compute stream only waits for the piece it is about to read.

`ShardPlan`, `split_phases`, `take_rows`, the exchanges and `PropagateEngine` are device- and backend-agnostic
(exercised with gloo on CPU in tests/test_sharding_gloo.py, the product kernels swapped for a torch
restatement); the operator builders and the layers at the bottom are the HIP compute path.
`EmulatedExchange` rehearses one rank of a P-rank job on a single GPU with the wire time of every exchange
played by a timed kernel on a separate stream (tools/emulate_sharded.py): the pipeline, its events and the
per-rank kernels are the production ones, only the bytes do not cross a link.
"""
import math
import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .sparse import CSR

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------------
# ownership
# ------------------------------------------------------------------------------------------------
def balanced_bounds(cost: Tensor, world_size: int) -> List[int]:
    """Contiguous ranges of (nearly) equal total `cost` (one entry per node): bounds[g] = first node whose
    prefix cost reaches g / P of the total.  SURVEY.md 8(e) "equal-nnz row ranges (not equal-N)"."""
    n = int(cost.numel())
    if n == 0:
        return [0] * (world_size + 1)
    prefix = torch.cumsum(cost.double().cpu(), 0)
    total = float(prefix[-1])
    targets = torch.tensor([total * g / world_size for g in range(1, world_size)], dtype=torch.float64)
    cuts = torch.searchsorted(prefix, targets, right=False).tolist()
    bounds = [0] + [min(int(c) + 1, n) for c in cuts] + [n]
    for g in range(1, len(bounds)):                      # monotone even for degenerate costs
        bounds[g] = max(bounds[g], bounds[g - 1])
    return bounds


def degree_cost(edge_index: Tensor, num_nodes: int) -> Tensor:
    """1 + in-degree + out-degree per node: the entries of a node's row of the symmetrised operator (reciprocal
    pairs counted twice -- a balance heuristic, not a count) plus its diagonal."""
    ones = torch.ones(edge_index.size(1), dtype=torch.float32, device=edge_index.device)
    deg = torch.ones(num_nodes, dtype=torch.float32, device=edge_index.device)
    deg.index_add_(0, edge_index[0], ones)
    deg.index_add_(0, edge_index[1], ones)
    return deg


class ShardPlan:
    """Contiguous node ranges (equal size by default, `bounds` for balanced ones) padded to n_pad rows each;
    n_pad is a multiple of `align` (the engine needs equal phases / row chunks)."""

    def __init__(self, num_nodes: int, world_size: int, rank: int, bounds: Optional[Sequence[int]] = None,
                 align: int = 1):
        if not (0 <= rank < world_size):
            raise ValueError(f"rank {rank} outside world of {world_size}")
        self.num_nodes, self.world_size, self.rank = int(num_nodes), int(world_size), int(rank)
        if bounds is None:
            step = (self.num_nodes + world_size - 1) // world_size
            bounds = [min(g * step, self.num_nodes) for g in range(world_size + 1)]
        bounds = [int(b) for b in bounds]
        if len(bounds) != world_size + 1 or bounds[0] != 0 or bounds[-1] != self.num_nodes or \
                any(bounds[g] > bounds[g + 1] for g in range(world_size)):
            raise ValueError(f"bounds {bounds} do not partition {self.num_nodes} nodes over {world_size} ranks")
        self.bounds = bounds
        self.sizes = [bounds[g + 1] - bounds[g] for g in range(world_size)]
        align = max(int(align), 1)
        self.n_pad = max((max(self.sizes) + align - 1) // align * align, align)
        self.n_total = self.n_pad * world_size          # padded node count (pad nodes are isolated)
        self.lo, self.hi = bounds[rank], bounds[rank + 1]
        self.n_local = self.hi - self.lo                # real rows owned by this rank
        self.pad_lo = rank * self.n_pad                 # first padded id of this rank
        self._index_cache = {}

    # ---- global <-> padded ids ---------------------------------------------------------------------
    def to_padded(self, ids: Tensor) -> Tensor:
        """Global node ids -> padded ids g * n_pad + (v - bounds[g])."""
        inner = torch.tensor(self.bounds[1:-1], dtype=ids.dtype, device=ids.device)
        ids = ids.contiguous()
        g = torch.searchsorted(inner, ids, right=True) if inner.numel() else torch.zeros_like(ids)
        starts = torch.tensor(self.bounds[:-1], dtype=ids.dtype, device=ids.device)
        return g * self.n_pad + ids - starts[g]

    def global_index(self, device) -> Tensor:
        """int64 [num_nodes]: padded id of every global node (to read a gathered [n_total, ...] buffer back in
        global order)."""
        key = str(device)
        if key not in self._index_cache:
            self._index_cache[key] = self.to_padded(torch.arange(self.num_nodes, dtype=torch.long, device=device))
        return self._index_cache[key]

    def shard_rows(self, x: Tensor) -> Tensor:
        """Rows of a global [num_nodes, F] matrix owned by this rank, zero-padded to n_pad rows."""
        out = x.new_zeros((self.n_pad,) + tuple(x.shape[1:]))
        if self.n_local > 0:
            out[:self.n_local] = x[self.lo:self.hi]
        return out

    def unshard_rows(self, gathered: Tensor) -> Tensor:
        """[n_total, ...] gathered buffer (rank-major) -> the [num_nodes, ...] global matrix."""
        return gathered.index_select(0, self.global_index(gathered.device))


# ------------------------------------------------------------------------------------------------
# exchanges
# ------------------------------------------------------------------------------------------------
class _Done:
    def wait(self):
        return True


class DistExchange:
    """torch.distributed collectives.  RCCL ("nccl"): asynchronous, on the process group's own stream;
    `handle.wait()` makes the CURRENT stream wait, so compute queued before the wait overlaps the transfer.
    gloo (tests; several ranks sharing one GPU): device tensors are staged through the host, synchronously."""
    emulated = False

    def __init__(self, group=None, synchronous: bool = False):
        self.group = group
        self.world_size, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self._staged = dist.get_backend(group) == "gloo"
        # synchronous: every collective is waited for where it is issued (no overlap) -- the conservative schedule
        # bench.py falls back to when the pipelined one fails on a machine
        self.synchronous = bool(synchronous)

    def _issue(self, work):
        if self.synchronous:
            work.wait()
            return _Done()
        return work

    @staticmethod
    def _wire(t: Tensor) -> Tensor:
        """gloo moves bytes: bf16 payloads travel as uint8 (gloo knows neither bf16 nor int16 everywhere)."""
        return t.view(torch.uint8) if t.dtype == torch.bfloat16 else t

    def _gather_sync(self, out: Tensor, inp: Tensor):
        flat = out.view((out.size(0) * out.size(1),) + tuple(out.shape[2:]))      # gloo wants the concatenated form
        try:
            dist.all_gather_into_tensor(flat, inp, group=self.group)
        except (RuntimeError, NotImplementedError):        # a backend without the fused form
            dist.all_gather(list(out.unbind(0)), inp, group=self.group)

    def all_gather(self, out: Tensor, inp: Tensor):
        """out [world, ...] <- inp [...] of every rank."""
        if self._staged:
            if inp.is_cuda:
                host = torch.empty(out.shape, dtype=out.dtype)
                self._gather_sync(self._wire(host), self._wire(inp.cpu().contiguous()))
                out.copy_(host)
            else:
                self._gather_sync(self._wire(out), self._wire(inp.contiguous()))
            return _Done()
        return self._issue(dist.all_gather_into_tensor(out, inp.contiguous(), group=self.group, async_op=True))

    def all_to_all(self, out: Tensor, inp: Tensor):
        """chunk d of inp [world, ...] goes to rank d; chunk s of out comes from rank s."""
        if self._staged:
            if inp.is_cuda:
                host = torch.empty(out.shape, dtype=out.dtype)
                dist.all_to_all_single(self._wire(host), self._wire(inp.cpu().contiguous()), group=self.group)
                out.copy_(host)
            else:
                dist.all_to_all_single(self._wire(out), self._wire(inp.contiguous()), group=self.group)
            return _Done()
        return self._issue(dist.all_to_all_single(out, inp.contiguous(), group=self.group, async_op=True))

    def all_reduce(self, t: Tensor) -> Tensor:
        if self.world_size > 1:
            if self._staged:
                host = t.detach().float().cpu()            # sums in fp32 whatever the storage type
                dist.all_reduce(host, group=self.group)
                t.copy_(host)
            else:
                dist.all_reduce(t, group=self.group)
        return t

    def all_reduce_async(self, t: Tensor):
        """Sum `t` over the ranks without holding up the compute stream: returns a handle whose wait() makes the
        current stream wait for the result (RCCL: the reduction runs on the process group's stream meanwhile)."""
        if self.world_size == 1 or self._staged:
            self.all_reduce(t)
            return _Done()
        return self._issue(dist.all_reduce(t, group=self.group, async_op=True))

    def broadcast_list(self, values: list, src: int = 0) -> list:
        box = [values]
        dist.broadcast_object_list(box, src=src, group=self.group)
        return box[0]


class ThreadExchange:
    """The ranks of a job as THREADS of one process that share one device (or the host): every collective is a
    rendezvous of the threads plus plain tensor copies on the one stream they all enqueue to -- correct values (unlike
    `EmulatedExchange`), no process group, no second device context.  `ThreadExchange.create(P)` returns the P
    exchanges; each thread builds its layer / engine with its own one and runs the same SPMD code as a real rank.
    For checking the sharded path at sizes where several PROCESSES on one GPU are impractical (eight device contexts
    on the one test GPU made a layer construction at the 20M-edge size take minutes, for reasons of the runtime's
    queue scheduling, not of this code) and for single-GPU rehearsals with real numbers.
    Autograd's engine runs every CUDA backward on ONE worker thread per device, so collectives inside a backward
    would wait for peers that can never run: drive the layers' Function.forward / .backward directly
    (tests/test_gpu_fullsize.py) or keep to forward-only use."""
    emulated = False
    synchronous = True

    class _Shared:
        def __init__(self, world_size):
            import threading
            self.barrier = threading.Barrier(world_size)
            self.slots = [None] * world_size

    def __init__(self, shared, world_size: int, rank: int):
        self._sh, self.world_size, self.rank, self.group = shared, int(world_size), int(rank), None

    @classmethod
    def create(cls, world_size: int):
        shared = cls._Shared(world_size)
        return [cls(shared, world_size, r) for r in range(world_size)]

    def _rendezvous(self, payload, consume):
        sh = self._sh
        sh.slots[self.rank] = payload
        sh.barrier.wait()                                # every payload is posted (and its producer enqueued)
        result = consume(sh.slots)                       # reads of the peers' payloads are enqueued here ...
        sh.barrier.wait()                                # ... before any owner may enqueue an overwrite
        return result

    def all_gather(self, out: Tensor, inp: Tensor):
        def consume(slots):
            for s, t in enumerate(slots):
                out[s].copy_(t)
        self._rendezvous(inp, consume)
        return _Done()

    def all_to_all(self, out: Tensor, inp: Tensor):
        def consume(slots):
            for s, t in enumerate(slots):
                out[s].copy_(t[self.rank])
        self._rendezvous(inp, consume)
        return _Done()

    def all_reduce(self, t: Tensor) -> Tensor:
        def consume(slots):
            total = slots[0].clone()                     # fixed order: every rank computes the same sum
            for other in slots[1:]:
                total = total + other
            return total
        total = self._rendezvous(t, consume)
        t.copy_(total)
        self._sh.barrier.wait()
        return t

    def all_reduce_async(self, t: Tensor):
        self.all_reduce(t)
        return _Done()

    def broadcast_list(self, values: list, src: int = 0) -> list:
        return self._rendezvous(values, lambda slots: list(slots[src]))


class _StreamEvent:
    def __init__(self, event):
        self.event = event

    def wait(self):
        from ._cabi import current_stream
        current_stream().wait_event(self.event)
        return True


class EmulatedExchange:
    """Rank `rank` of a `world_size`-rank job rehearsed on ONE GPU: every exchange is played on a separate HIP
    stream as (a device copy of the bytes this rank would receive) + (a timed kernel of the wire time of the
    busiest link, bytes_per_link / link_gbps + latency: one idle lane polling the wall clock).  The received
    VALUES are this rank's own data repeated -- the rehearsal measures the pipeline (streams, events, per-rank
    kernels at their real sizes), it does not compute a correct product.  Two variants were tried and dropped
    because they disturbed what they were meant to measure (profiles/r2_emulated_sharded_w8.json keeps the
    default's numbers, reproduced in a later run): a high-priority side stream (compute kernels ran 2x slower in
    about a third of the shapes) and spending the wire time on the copy engine (device -> pinned host transfers
    serialised against the compute stream).  Known artefact of the default: when the runtime maps the side stream
    to the hardware queue of the compute stream, the timed kernel sits in front of compute work and the wire time
    shows up as packing time (seen in the row layout with two phases); those rows are left in the files."""
    emulated = True

    def __init__(self, world_size: int, rank: int, link_gbps: float = 61.0, latency_us: float = 10.0):
        from . import _cabi
        self._cabi = _cabi
        self.world_size, self.rank = int(world_size), int(rank)
        self.link_gbps, self.latency_us = float(link_gbps), float(latency_us)
        self.stream = torch.cuda.Stream()
        self.group = None
        self.wire_us = 0.0                               # accumulated emulated wire time (reset by the caller)

    def _play(self, out: Tensor, src: Tensor, bytes_per_link: float):
        ready = torch.cuda.Event()
        ready.record(self._cabi.current_stream())
        us = self.latency_us + bytes_per_link / (self.link_gbps * 1e3) if self.world_size > 1 else 0.0
        self.wire_us += us
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            out.copy_(src)
            if us > 0:
                self._cabi.check(self._cabi.lib().pygsd_spin_us(us, self._cabi.stream_ptr()), "pygsd_spin_us")
            done = torch.cuda.Event()
            done.record(self.stream)
        return _StreamEvent(done)        # buffers are the engine's persistent ones: nothing to keep alive

    def all_gather(self, out: Tensor, inp: Tensor):
        return self._play(out, inp.unsqueeze(0).expand_as(out), inp.numel() * inp.element_size())

    def all_to_all(self, out: Tensor, inp: Tensor):
        return self._play(out, inp, inp[0].numel() * inp.element_size())

    def all_reduce(self, t: Tensor) -> Tensor:
        return t

    def all_reduce_async(self, t: Tensor):
        return _Done()

    def broadcast_list(self, values: list, src: int = 0) -> list:
        return values


# ------------------------------------------------------------------------------------------------
# CSR surgery (pure index arithmetic; torch ops on whatever device the CSR lives on)
# ------------------------------------------------------------------------------------------------
def _row_of_slot(rowptr: Tensor, nnz: int) -> Tensor:
    counts = (rowptr[1:] - rowptr[:-1]).long()
    return torch.repeat_interleave(torch.arange(counts.numel(), dtype=torch.long, device=rowptr.device), counts,
                                   output_size=nnz)


def take_rows(csr: CSR, values: Sequence[Tensor], row_ids: Tensor) -> Tuple[CSR, Tuple[Tensor, ...]]:
    """The sub-operator made of rows `row_ids` (int64, any order, no repeats needed) in THAT order; columns
    untouched.  A row's entries keep their order."""
    row_ids = row_ids.long()
    rp = csr.rowptr.long()
    counts = rp[row_ids + 1] - rp[row_ids]
    new_ptr = torch.zeros(row_ids.numel() + 1, dtype=torch.long, device=rp.device)
    new_ptr[1:] = torch.cumsum(counts, 0)
    nnz = int(new_ptr[-1]) if row_ids.numel() else 0
    owner = torch.repeat_interleave(torch.arange(row_ids.numel(), dtype=torch.long, device=rp.device), counts,
                                    output_size=nnz)
    src = torch.arange(nnz, dtype=torch.long, device=rp.device) - new_ptr[owner] + rp[row_ids][owner]
    out = CSR(int(row_ids.numel()), csr.n_cols, nnz, new_ptr.to(torch.int32), csr.col[src].contiguous(), None)
    return out, tuple(v[src].contiguous() for v in values)


def cut_points(total: int, count: int, fracs: Optional[Sequence[float]] = None) -> List[int]:
    """count + 1 ascending bounds of [0, total): equal pieces (total must divide) or -- `fracs`, positive, any sum -- pieces
    in those proportions, every piece at least one row while total >= count (an exchange of zero rows is no exchange).  With
    fewer rows than pieces (a shard of a toy graph) some pieces ARE empty: their exchanges move nothing and their products
    launch nothing (the kernels return on zero rows; tests/test_sharding_gloo.py runs such a plan)."""
    if fracs is None:
        if total % count:
            raise ValueError(f"{total} rows do not split into {count} equal pieces")
        step = total // count
        return [k * step for k in range(count)] + [total]
    if len(fracs) != count or any(f <= 0 for f in fracs):
        raise ValueError(f"{count} positive fractions expected, got {tuple(fracs)}")
    scale, acc, out = float(sum(fracs)), 0.0, [0]
    for k, f in enumerate(fracs[:-1]):
        acc += f / scale
        lo = out[-1] + (1 if total >= count else 0)
        hi = total - (count - 1 - k if total >= count else 0)
        out.append(min(max(int(round(acc * total)), lo), hi))
    return out + [total]


def split_spec(spec) -> Tuple[int, Optional[Tuple[float, ...]]]:
    """A pipeline depth given as a count (equal pieces) or as a sequence of fractions (uneven pieces: a short FIRST inbound
    phase puts the first product on the compute stream early, a short LAST return chunk leaves little behind the last
    product) -> (count, fractions or None).  A string "2" / "0.4,0.6" (environment) is parsed the same way."""
    if isinstance(spec, str):
        parts = [t for t in spec.replace(":", ",").split(",") if t.strip()]
        if len(parts) == 1:                  # one number is a COUNT ("2", "2.0"); "0.5" alone is neither a count nor a split
            value = float(parts[0])
            if value != int(value) or value < 1:
                raise ValueError(f"pipeline depth {spec!r}: a count (\"2\") or at least two fractions (\"0.4,0.6\")")
            spec = int(value)
        else:
            spec = [float(t) for t in parts]
    if isinstance(spec, (int,)) or (hasattr(spec, "__int__") and not hasattr(spec, "__len__")):
        return max(int(spec), 1), None
    fr = tuple(float(f) for f in spec)
    if len(fr) == 1:
        return 1, None
    return len(fr), fr


def split_phases(csr: CSR, values: Sequence[Tensor], n_pad: int, phases: int, world_size: int,
                 bounds: Optional[Sequence[int]] = None) -> List[Tuple[CSR, Tuple[Tensor, ...]]]:
    """Column blocks of an operator whose columns are PADDED node ids: block c holds the entries whose column
    falls in sub-range c = [bounds[c], bounds[c + 1]) of ITS rank's range (`phases` equal pieces without `bounds`), with
    the column re-based into phase c's exchange buffer [world, n_sub_c, ...] viewed as rows:
    (col // n_pad) * n_sub_c + (col % n_pad) - bounds[c].
    A row's entries keep their order inside a block.  The blocks' col / value arrays are VIEWS into one buffer sorted
    by (phase, row): they keep all phases' memory alive together and are 4-byte aligned only."""
    if phases == 1:
        return [(csr, tuple(values))]
    if bounds is None:
        if n_pad % phases:
            raise ValueError(f"n_pad = {n_pad} is not a multiple of {phases} phases")
        bounds = cut_points(n_pad, phases)
    bounds = [int(b) for b in bounds]
    if len(bounds) != phases + 1 or bounds[0] != 0 or bounds[-1] != n_pad or any(a > b for a, b in zip(bounds, bounds[1:])):
        raise ValueError(f"phase bounds {bounds} do not partition {n_pad} rows into {phases} phases")
    n_subs = [bounds[c + 1] - bounds[c] for c in range(phases)]
    n_rows, nnz = csr.n_rows, csr.nnz
    if n_rows == 0 or nnz == 0:                    # an empty shard: every phase is an empty block of the same shape
        empty_ptr = torch.zeros(n_rows + 1, dtype=torch.int32, device=csr.rowptr.device)
        return [(CSR(n_rows, world_size * n_subs[c], 0, empty_ptr, csr.col[:0], None), tuple(v[:0] for v in values))
                for c in range(phases)]
    col = csr.col.long()
    local = col % n_pad
    inner = torch.tensor(bounds[1:-1], dtype=torch.long, device=col.device)
    phase = torch.searchsorted(inner, local, right=True)
    lo_t = torch.tensor(bounds[:-1], dtype=torch.long, device=col.device)
    sub_t = torch.tensor(n_subs, dtype=torch.long, device=col.device)
    compact = ((col // n_pad) * sub_t[phase] + local - lo_t[phase]).to(torch.int32)
    # ONE stable sort by (phase, row) instead of a boolean mask per phase: a handful of launches and a single
    # device -> host read whatever the number of phases (mask indexing costs a host round trip per mask and array;
    # with several ranks sharing one GPU those round trips took minutes at the 20M-edge size)
    key = phase * n_rows + _row_of_slot(csr.rowptr, nnz)
    skey, order = torch.sort(key, stable=True)
    edges = torch.arange(phases * n_rows + 1, dtype=torch.long, device=col.device)
    ptr_all = torch.searchsorted(skey, edges)                      # first sorted entry of every (phase, row)
    starts = ptr_all[::n_rows].tolist()                            # phase boundaries: the one host read
    cols = compact[order]
    vals = [v[order] for v in values]
    out = []
    for c in range(phases):
        lo, hi = int(starts[c]), int(starts[c + 1])
        new_ptr = (ptr_all[c * n_rows:(c + 1) * n_rows + 1] - lo).to(torch.int32)
        sub = CSR(n_rows, world_size * n_subs[c], hi - lo, new_ptr, cols[lo:hi], None)
        out.append((sub, tuple(v[lo:hi] for v in vals)))
    return out


class PhasedOperator:
    """One orientation of the operator rows a rank multiplies: per phase a CSR (columns = rows of that phase's
    exchange buffer) and its value arrays.  dual: two value arrays on ONE pattern, applied to feature groups
    0 and 1 in one traversal (magnetic real / imaginary parts)."""

    def __init__(self, blocks: List[Tuple[CSR, Tuple[Tensor, ...]]], dual: bool, mean: bool = False):
        self.blocks, self.dual, self.mean = blocks, dual, mean
        self.n_rows = blocks[0][0].n_rows
        self.nnz = sum(b[0].nnz for b in blocks)
        if mean and len(blocks) > 1:
            raise ValueError("a mean-reduced operator cannot be split into column blocks")


# ------------------------------------------------------------------------------------------------
# product kernels (HIP); the CPU tests pass torch restatements with the same signatures
# ------------------------------------------------------------------------------------------------
def _hip_dual(csr, va, vb, xa, xb, ya, yb, lo, hi, alpha, accumulate):
    from .sparse import spmm2_rows_into
    spmm2_rows_into(csr, va, vb, xa, xb, ya, yb, lo, hi, alpha, accumulate)


def _hip_single(csr, val, x, y, lo, hi, alpha, accumulate, mean):
    from .sparse import spmm_rows_into
    spmm_rows_into(csr, val, x, y, lo, hi, alpha, accumulate, mean)


class PropagateEngine:
    """Executes  Y_g = alpha * S_g X_g  (g = feature group) for the operator rows of one rank: exchanges in,
    pipelined partial products, exchange back (grid only).  See the module docstring for the schedule."""

    def __init__(self, plan: ShardPlan, exchange, p_c: int = 1, phases=1, return_chunks=1,
                 kernels: Optional[Tuple[Callable, Callable]] = None, force_grid: bool = False):
        self.plan, self.ex = plan, exchange
        world = plan.world_size
        if world % p_c:
            raise ValueError(f"{p_c} column slices do not divide {world} ranks")
        self.p_c, self.p_r = int(p_c), world // int(p_c)
        # force_grid: the grid schedule (all-to-all in, row-chunked all-to-all back, merge) with ONE column slice --
        # the degenerate 1 x 1 / p_r x 1 grid, so that a single rank can run the whole schedule over RCCL
        self.grid = self.p_c > 1 or bool(force_grid)
        self.i, self.j = plan.rank // self.p_c, plan.rank % self.p_c
        # phases / return_chunks: a count (equal pieces) or a sequence of fractions (round 5: uneven pieces -- the compute
        # stream idles for the FIRST inbound phase and for the LAST return chunk only, so those two are the ones to keep short)
        self.phases, self.phase_fracs = split_spec(phases)
        self.return_chunks, self.chunk_fracs = split_spec(return_chunks) if self.grid else (1, None)
        if (self.phase_fracs is None and plan.n_pad % self.phases) or (self.grid and plan.n_pad % self.p_r) or \
                (self.grid and self.chunk_fracs is None and plan.n_pad % (self.p_r * self.return_chunks)):
            raise ValueError(f"n_pad = {plan.n_pad} must be a multiple of the (equal) phases ({self.phases}) and of "
                             f"p_r x (equal) return chunks ({self.p_r} x {self.return_chunks}); build the plan with "
                             f"align = PropagateEngine.alignment(...)")
        self.n_blk = plan.n_pad // self.p_r if self.grid else plan.n_pad
        self.phase_bounds = cut_points(plan.n_pad, self.phases, self.phase_fracs)          # inside a rank's range
        self.chunk_bounds = cut_points(self.n_blk, self.return_chunks, self.chunk_fracs)   # inside a row block
        self.phase_rows = [b - a for a, b in zip(self.phase_bounds, self.phase_bounds[1:])]
        self.chunk_rows = [b - a for a, b in zip(self.chunk_bounds, self.chunk_bounds[1:])]
        self.n_sub = self.phase_rows[0] if self.phase_fracs is None else None     # (equal pieces only: the old scalar names)
        self.n_rsub = self.chunk_rows[0] if self.chunk_fracs is None else None
        self.block_rows = world * self.n_blk if self.grid else plan.n_pad
        self.dual_kernel, self.single_kernel = kernels or (_hip_dual, _hip_single)
        self.timing = None                                   # dict of event lists when profiling
        self._scratch = {}

    def _buf(self, name: str, shape, like: Tensor) -> Tensor:
        """Exchange / staging buffers live as long as the engine: a propagate never allocates for them (a
        buffer handed to a collective on another stream would otherwise pin its allocator block until that
        stream's event retires, and the next propagate would fall through to hipMalloc).  Reuse is safe: the
        compute stream has waited for every exchange of the previous propagate before it packs the next one."""
        key = (name, tuple(shape), like.dtype, like.device)
        t = self._scratch.get(key)
        if t is None:
            # zero-filled once: the pad rows of a send buffer are never written by the dense backward's packed epilogue (it runs
            # on the real rows only) and must read as zero gradient rows
            t = self._scratch[key] = torch.zeros(tuple(shape), dtype=like.dtype, device=like.device)
        return t

    # ---- piece layouts (round 5): the dense kernels write / read the exchange buffers directly ------
    def pieces_ok(self, f: int, like: Tensor, groups: int = 2) -> bool:
        """Can the dense kernels address this engine's exchange buffers (include/pygsd_hip.h: pygsd_piece_layout)?  fp32 on the
        GPU, column slices of 16 * 2^k floats, at most 4 phases / return chunks and 8 row blocks, and every slot (rows of a
        phase / chunk x the `groups` feature groups side by side) below 2^31 floats -- piece_layout_check's bound; callers fall
        back to the tensor-op pack / merge beyond it."""
        if not (like.is_cuda and like.dtype == torch.float32) or f % self.p_c:
            return False
        fw = f // self.p_c
        pieces = fw // 16
        return (fw % 16 == 0 and pieces & (pieces - 1) == 0 and self.phases <= 4 and self.return_chunks <= 4 and self.p_r <= 8
                and max(self.phase_rows + self.chunk_rows) * max(groups, 2) * fw < (1 << 31))

    def send_layout(self, groups: int, f: int, like: Tensor):
        """(layout, send buffers of all phases): where a local [n_pad, f] operand of `groups` feature groups goes in the inbound
        exchange's send buffers -- what `_pack_phase` produces, as addresses.  Group g's operand pointer is buffer 0 + g * fw."""
        from ._cabi import PieceLayout
        p_r, p_c = (self.p_r, self.p_c) if self.grid else (1, 1)
        fw = f // p_c
        lead = (self.plan.world_size,) if self.grid else ()
        bufs = [self._buf(f"send{c}", lead + (self.phase_rows[c], groups * fw), like) for c in range(self.phases)]
        esz = like.element_size()
        base = [(b.data_ptr() - bufs[0].data_ptr()) // esz for b in bufs]
        return PieceLayout(base, self.phase_bounds, self.phase_rows, self.plan.n_pad, p_c, groups * fw, fw, p_r), bufs

    def return_layout(self, groups: int, fw: int):
        """Where row t (local order) of a returned product lives in the flat receive buffer of the return exchange
        [block_rows, groups * fw] -- what `_merge` reads, as addresses (grid only)."""
        from ._cabi import PieceLayout
        world = self.plan.world_size
        base = [world * self.chunk_bounds[r] * groups * fw for r in range(self.return_chunks)]
        return PieceLayout(base, self.chunk_bounds, self.chunk_rows, self.n_blk, self.p_c, groups * fw, fw, 1)

    @staticmethod
    def alignment(world_size: int, p_c: int, phases, return_chunks, force_grid: bool = False) -> int:
        p_r = world_size // p_c
        n_ph, ph_fr = split_spec(phases)
        n_rc, rc_fr = split_spec(return_chunks)
        a = n_ph if ph_fr is None else 1                      # uneven pieces need no divisibility
        b = (p_r * (n_rc if rc_fr is None else 1)) if (p_c > 1 or force_grid) else 1
        return a * b // math.gcd(a, b)

    # ---- which operator rows this rank multiplies, in product order ---------------------------------
    def block_row_ids(self, device) -> Tensor:
        """Padded ids of the rows of this rank's products.  rows layout: the own range.  grid: row block i =
        the i-th 1/p_r of every rank's range, ordered (return chunk r, owner g, row in chunk) so that return
        chunk r is a contiguous row range of the product and an equal-split all-to-all."""
        plan = self.plan
        if not self.grid:
            return torch.arange(plan.pad_lo, plan.pad_lo + plan.n_pad, dtype=torch.long, device=device)
        g = torch.arange(plan.world_size, dtype=torch.long, device=device).view(-1, 1)
        ids = []
        for r in range(self.return_chunks):
            t = torch.arange(self.chunk_bounds[r], self.chunk_bounds[r + 1], dtype=torch.long, device=device).view(1, -1)
            ids.append((g * plan.n_pad + self.i * self.n_blk + t).reshape(-1))
        return torch.cat(ids)

    def phased(self, csr: CSR, values: Sequence[Tensor], dual: bool, mean: bool = False) -> PhasedOperator:
        """Operator rows (already restricted / ordered by `block_row_ids`, padded column ids) -> PhasedOperator."""
        return PhasedOperator(split_phases(csr, values, self.plan.n_pad, self.phases, self.plan.world_size,
                                           self.phase_bounds), dual, mean)

    # ---- packing ------------------------------------------------------------------------------------
    def _pack(self, xs: Sequence[Tensor], c: int) -> Tensor:
        """Sub-range c of the local rows of every feature group, packed for the exchange.
        rows: [n_sub, G * F] (groups side by side).  grid: [world, n_sub, G * fw], chunk d = column slice d % p_c."""
        rows, n_sub = slice(self.phase_bounds[c], self.phase_bounds[c + 1]), self.phase_rows[c]
        if not self.grid:
            f = xs[0].size(1)
            out = self._buf(f"send{c}", (n_sub, len(xs) * f), xs[0])
            for g, x in enumerate(xs):
                out[:, g * f:(g + 1) * f] = x[rows]
            return out
        fw, groups = xs[0].size(1) // self.p_c, len(xs)
        out = self._buf(f"send{c}", (self.plan.world_size, n_sub, groups * fw), xs[0])
        dst = out.view(self.p_r, self.p_c, n_sub, groups, fw)
        for g, x in enumerate(xs):          # one strided copy per group writes all p_r replicas of the p_c slices
            dst[:, :, :, g, :] = x[rows].reshape(n_sub, self.p_c, fw).permute(1, 0, 2)
        return out

    def _pack_phase(self, xs: Sequence[Tensor], c: int) -> Tensor:
        """The send buffer of phase c.  On the GPU: one launch of pygsd_pack_slices per phase (every 16-byte unit of
        the inputs read once, written p_r times; per phase, so that phase 0 is on the wire while phase 1 is packed);
        elsewhere the tensor-op restatement `_pack`."""
        if not xs[0].is_cuda:
            return self._pack(xs, c)
        from . import _cabi
        ld = xs[0].stride(0)
        if any(x.stride(1) != 1 or x.stride(0) != ld for x in xs):
            raise ValueError("feature groups of one propagate must be row-major with one common row stride")
        f, groups, esz = xs[0].size(1), len(xs), xs[0].element_size()
        p_r, p_c = (self.p_r, self.p_c) if self.grid else (1, 1)
        lead = (self.plan.world_size,) if self.grid else ()
        n_sub = self.phase_rows[c]
        out = self._buf(f"send{c}", lead + (n_sub, groups * (f // p_c)), xs[0])
        first = self.phase_bounds[c] * ld * esz
        ptrs = (_cabi.c_void_p * groups)(*[x.data_ptr() + first for x in xs])
        with _cabi.on_device(xs[0].device):
            _cabi.check(_cabi.lib().pygsd_pack_slices(ptrs, groups, n_sub, f * esz, ld * esz, p_r, p_c, 1,
                                                      _cabi.ptr(out), _cabi.stream_ptr()), "pygsd_pack_slices")
        return out

    def chunk_view(self, flat: Tensor, r: int) -> Tensor:
        """Return chunk r of a flat [block_rows, W] product / receive buffer as the all-to-all operand [world, rows of chunk r, W]
        (the chunks are consecutive row ranges of world x chunk_rows[r] rows)."""
        world = self.plan.world_size
        lo = world * self.chunk_bounds[r]
        return flat[lo:lo + world * self.chunk_rows[r]].view(world, self.chunk_rows[r], flat.size(-1))

    def _merge(self, recv: Tensor, groups: int) -> List[Tensor]:
        """recv: flat [block_rows, G * fw]; its chunk r (`chunk_view`) holds, from rank s = i' * p_c + j', the rows
        (block i', chunk r) of MY range x column slice j' -> per group the [n_pad, F] rows in local order
        i' * n_blk + chunk_bounds[r] + t."""
        fw = recv.size(-1) // groups
        if groups <= 4 and self.pieces_ok(self.p_c * fw, recv, groups):
            # one HIP pass whatever the chunk sizes (pygsd_gather_pieces_f32)
            from .dense import PieceOperand, gather_pieces
            return gather_pieces(PieceOperand(recv, 0, fw, self.return_layout(groups, fw)), self.plan.n_pad, self.p_c * fw,
                                 groups=groups)
        if self.chunk_fracs is None:
            v = recv.view(self.return_chunks, self.p_r, self.p_c, self.n_rsub, groups, fw)
            v = v.permute(4, 1, 0, 3, 2, 5).reshape(groups, self.plan.n_pad, self.p_c * fw)
            return [v[g] for g in range(groups)]
        out = recv.new_empty((groups, self.p_r, self.n_blk, self.p_c, fw))
        for r in range(self.return_chunks):
            v = self.chunk_view(recv, r).view(self.p_r, self.p_c, self.chunk_rows[r], groups, fw)
            out[:, :, self.chunk_bounds[r]:self.chunk_bounds[r + 1]] = v.permute(3, 0, 2, 1, 4)
        out = out.view(groups, self.plan.n_pad, self.p_c * fw)
        return [out[g] for g in range(groups)]

    # ---- the propagate ------------------------------------------------------------------------------
    def run(self, xs: Optional[Sequence[Tensor]], op, alpha: float = 1.0, prepacked=None, merge: bool = True,
            input_memo=None, whole_op=None):
        """xs: G local [n_pad, F] feature groups.  op: a dual PhasedOperator (G = 2) or a list of G single ones.
        Returns G local [n_pad, F] products.
        input_memo (a memo.TensorMemo, opt-in): the INBOUND exchange of these very tensors -- same objects, same in-place
        version -- is not repeated: the received buffers of the last propagate over them are read again.  For operands that do
        not change between steps (the input features of a first layer on a fixed graph); never for gradients.  whole_op: the same
        operator rows as `op` in ONE column block (padded ids); with nothing on the wire there is nothing to overlap, so a memo hit
        multiplies them in a single walk over the kept rows (the phases' received pieces joined once, when they were kept).
        prepacked = (G, F, like) with xs = None: the send buffers already hold the operand (`send_layout`: the dense backward
        wrote it there) -- no packing pass.  merge = False (grid): the products stay where the return exchange put them -- a FRESH
        receive buffer, handed back as a dense.PieceOperand whose layout the dense kernels / pygsd_gather_pieces_f32 read."""
        plan, world = self.plan, self.plan.world_size
        if prepacked is not None:
            groups, f, like = prepacked
            xs = None
        else:
            groups, f, like = len(xs), xs[0].size(1), xs[0]
        dual = isinstance(op, PhasedOperator)
        if dual and (not op.dual or groups != 2):
            raise ValueError("a dual operator takes exactly two feature groups")
        if not merge and not self.grid:
            raise ValueError("merge = False: only the grid layout has a return exchange to read in place")
        quantum = self.p_c * (8 if like.dtype == torch.bfloat16 else 4)
        if xs is not None and f % quantum and xs[0].is_cuda:
            # the vector kernels address 16-byte row pieces: zero-pad odd widths (e.g. a 6-wide layer), slice after
            pad = (0, quantum - f % quantum)
            out = self.run([torch.nn.functional.pad(x, pad) for x in xs], op, alpha)
            return [y[:, :f] for y in out]
        if self.grid and f % self.p_c:
            raise ValueError(f"width {f} does not split into {self.p_c} column slices")
        fw = f // self.p_c
        ev = self._events()
        self._mark(ev, "start")
        works, bufs = [], []
        if xs is not None and xs[0].is_cuda:
            ld = xs[0].stride(0)
            if any(x.stride(1) != 1 or x.stride(0) != ld for x in xs):
                xs = [x.contiguous() for x in xs]
        sends = self.send_layout(groups, f, like)[1] if xs is None else None
        kept = None
        if input_memo is not None and xs is not None:
            kept = input_memo.get(tuple(xs), ("inbound", groups, f))
        n_phases, phase_rows = self.phases, self.phase_rows
        if kept is not None:                                 # the same tensors at the same version: what arrived then is still right
            bufs, works = list(kept), [_Done()] * len(kept)
            if len(kept) == 1 and self.phases > 1:           # (joined into one [world, n_pad, W] buffer: one walk with whole_op)
                op, n_phases, phase_rows = whole_op, 1, [self.plan.n_pad]
        else:
            for c in range(self.phases):                     # every exchange is issued before any product
                send = sends[c] if xs is None else self._pack_phase(xs, c)
                shape = (world, self.phase_rows[c], groups * fw)
                # (a memoised exchange keeps buffers of its own: the engine's are overwritten by the next propagate)
                buf = self._buf(f"recv{c}", shape, send) if input_memo is None or xs is None else send.new_empty(shape)
                works.append(self.ex.all_to_all(buf, send) if self.grid else self.ex.all_gather(buf, send))
                bufs.append(buf)
        self._mark(ev, "packed")
        # the row layout hands its products to the caller (fresh tensors).  The grid's products are written by the
        # SpMM STRAIGHT INTO the return exchange's send buffer: group g = columns [g fw, (g + 1) fw) of rows that are
        # already ordered (return chunk, owner, row) -- chunk r of `home` is the all-to-all input as it stands
        returns, recv, home = [], None, None
        if self.grid:
            home = self._buf("home", (self.block_rows, groups * fw), like)
            ys = [home[:, g * fw:(g + 1) * fw] for g in range(groups)]
            # read in place by the consumer: a buffer of its own (the forward's is kept for the backward pass)
            recv = self._buf("back", (self.block_rows, groups * fw), like) if merge else \
                like.new_empty((self.block_rows, groups * fw))
        else:
            # bf16 storage with more than one phase: the partial products accumulate in fp32 and are rounded ONCE
            widen = like.dtype == torch.bfloat16 and self.phases > 1
            ys = [like.new_empty((self.block_rows, fw), dtype=torch.float32 if widen else like.dtype)
                  for _ in range(groups)]
        for c in range(n_phases):
            works[c].wait()
            self._mark(ev, "arrived")
            buf = bufs[c].view(world * phase_rows[c], groups * fw)
            last = c == n_phases - 1
            spans = [(world * self.chunk_bounds[r], world * self.chunk_bounds[r + 1]) for r in range(self.return_chunks)] \
                if (last and self.grid) else [(0, self.block_rows)]
            for r, (lo, hi) in enumerate(spans):
                if dual:
                    csr, (va, vb) = op.blocks[c]
                    self.dual_kernel(csr, va, vb, buf[:, :fw], buf[:, fw:], ys[0], ys[1], lo, hi, alpha, c > 0)
                else:
                    for g in range(groups):
                        csr, (val,) = op[g].blocks[c]
                        self.single_kernel(csr, val, buf[:, g * fw:(g + 1) * fw], ys[g], lo, hi, alpha, c > 0,
                                           op[g].mean)
                if last and self.grid:                       # chunk r goes home while chunk r + 1 is multiplied
                    returns.append(self.ex.all_to_all(self.chunk_view(recv, r), self.chunk_view(home, r)))
            self._mark(ev, "multiplied")
        if input_memo is not None and xs is not None and kept is None:
            # keep what arrived (every piece has been waited for above) -- joined into ONE buffer when the caller has the operator
            # in one column block, so that the next propagate over these tensors is a single walk
            keep = tuple(bufs)
            if whole_op is not None and self.phases > 1:
                whole = like.new_empty((world, plan.n_pad, groups * fw))
                for c in range(self.phases):
                    whole[:, self.phase_bounds[c]:self.phase_bounds[c + 1]] = bufs[c]
                keep = (whole,)
            input_memo.put(tuple(xs), ("inbound", groups, f), keep)
        if not self.grid:
            if ys[0].dtype != like.dtype:
                ys = [y.to(like.dtype) for y in ys]
            self._mark(ev, "end")
            return ys
        for w in returns:
            w.wait()
        self._mark(ev, "returned")
        if not merge:
            from .dense import PieceOperand
            self._mark(ev, "end")
            return PieceOperand(recv, 0, fw, self.return_layout(groups, fw))
        out = self._merge(recv, groups)
        self._mark(ev, "end")
        return out

    # ---- instrumentation (bench.py / tools/emulate_sharded.py) ------------------------------------
    def profile(self, on: bool = True):
        self.timing = [] if on else None

    def _events(self):
        if self.timing is None:
            return None
        self.timing.append([])
        return self.timing[-1]

    def remark_end(self):
        """Profiling only: move the "end" mark of the last propagate behind what the caller has queued since (the fused merge of a
        product that `run(..., merge=False)` left in the receive buffer), so that `merge_ms` keeps meaning "from the last arrival
        to rows in local order" whoever runs the merge."""
        if self.timing and self.timing[-1] and self.timing[-1][-1][0] == "end":
            self.timing[-1].pop()
            self._mark(self.timing[-1], "end")

    @staticmethod
    def _mark(ev, name):
        if ev is not None and torch.cuda.is_available():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev.append((name, e))

    def timing_summary(self) -> Optional[dict]:
        """Per-propagate averages over the profiled runs (milliseconds on the compute stream):
        total, pack, wait_in (compute stream stalled on an inbound exchange), product, wait_out + merge."""
        if not self.timing:
            return None
        torch.cuda.synchronize()
        acc = {"total_ms": 0.0, "pack_ms": 0.0, "wait_in_ms": 0.0, "product_ms": 0.0, "wait_out_ms": 0.0, "merge_ms": 0.0}
        runs = 0
        for ev in self.timing:
            if not ev or ev[-1][0] != "end":
                continue
            runs += 1
            acc["total_ms"] += ev[0][1].elapsed_time(ev[-1][1])
            for (na, ea), (nb, eb) in zip(ev[:-1], ev[1:]):
                dt = ea.elapsed_time(eb)
                key = {"packed": "pack_ms", "arrived": "wait_in_ms", "multiplied": "product_ms",
                       "returned": "wait_out_ms", "end": "merge_ms"}[nb]
                acc[key] += dt
        if not runs:
            return None
        out = {k: v / runs for k, v in acc.items()}
        out["propagates"] = runs
        out["exposed_exchange_ms"] = out["wait_in_ms"] + out["wait_out_ms"]
        return out


# ------------------------------------------------------------------------------------------------
# layout choice
# ------------------------------------------------------------------------------------------------
def choose_cols(world_size: int, n_feat: int) -> int:
    """Largest p_c <= 4 dividing the world with 16-byte-aligned column slices (fw % 4 == 0): two packed slices
    of >= 16 floats still fill the 128-byte line a gather costs anyway.  Received volume relative to one full
    feature pair: rows (1 - 1/P); grid (1 - 1/P) / p_c in + (1 - 1/P) / p_c back.  A 1 x 2 grid on two ranks
    moves exactly what the all-gather moves, so two ranks stay in the row layout."""
    for p_c in (4, 2):
        if world_size % p_c == 0 and world_size > p_c - 1 + (p_c == 2) and n_feat % (4 * p_c) == 0:
            return p_c
    return 1


_MERGE_ON_READ = os.environ.get("PYGSD_SHARD_MERGE_ON_READ", "1") != "0"
_PACKED_BACKWARD = os.environ.get("PYGSD_SHARD_PACKED_BACKWARD", "1") != "0"


def set_exchange_shortcuts(merge_on_read: Optional[bool] = None, packed_backward: Optional[bool] = None):
    """Round 5's two shortcuts of the sharded magnetic layers (A / B runs, tests): consumers read returned products in place /
    the dense backward writes the last gradient term into the send buffers.  -> the previous (merge_on_read, packed_backward)."""
    global _MERGE_ON_READ, _PACKED_BACKWARD
    prev = (_MERGE_ON_READ, _PACKED_BACKWARD)
    if merge_on_read is not None:
        _MERGE_ON_READ = bool(merge_on_read)
    if packed_backward is not None:
        _PACKED_BACKWARD = bool(packed_backward)
    return prev


def _env_spec(name: str, default):
    """A pipeline depth from the environment: a count ("2") or fractions ("0.4,0.6")."""
    raw = os.environ.get(name)
    if not raw:
        return default
    try:
        count, fracs = split_spec(raw)
    except ValueError:
        return default
    return fracs if fracs is not None else count


# Round 5: uneven pieces.  The compute stream idles while the FIRST inbound phase is on the wire: a first phase of ~40 % of the rows
# is the shortest one whose product still covers the second phase's wire time (product ~ wire per entry on this workload) --
# rehearsed: 2.04 -> 2.00 ms per step at 8 ranks.  The return does NOT gain from finer or uneven chunks: every exchange carries a
# fixed cost (issue + arrival latency, ~0.03 - 0.05 ms in the rehearsal) and the return's total wire time is about the last phase's
# product, so what counts is how early the first chunk leaves, not how short the last one is -- three and four chunks measured
# 2.06 - 2.14 ms against 2.00 - 2.04 for two equal ones (profiles/r5b_emulated_w8.json).
DEFAULT_PHASES = (0.4, 0.6)
DEFAULT_RETURN_CHUNKS = 2


def default_pipeline(world_size: int, grid: bool):
    """(phases, return chunks) when there is an exchange to hide: counts or fraction tuples (PYGSD_SHARD_PHASES /
    PYGSD_SHARD_RETURN_CHUNKS override: "2" = two equal pieces, "0.4,0.6" = uneven ones)."""
    if world_size == 1:
        return 1, 1
    return _env_spec("PYGSD_SHARD_PHASES", DEFAULT_PHASES), (_env_spec("PYGSD_SHARD_RETURN_CHUNKS", DEFAULT_RETURN_CHUNKS)
                                                             if grid else 1)


def make_plan(num_nodes: int, exchange, edge_index: Optional[Tensor], p_c: int, phases, return_chunks,
              balance: bool = True, force_grid: bool = False) -> ShardPlan:
    """Equal-work ranges from the edge list (identical on every rank: rank 0's bounds are broadcast)."""
    world, rank = exchange.world_size, exchange.rank
    bounds = None
    if balance and edge_index is not None and world > 1:
        bounds = balanced_bounds(degree_cost(edge_index, num_nodes), world)
        bounds = exchange.broadcast_list(bounds, 0)
    return ShardPlan(num_nodes, world, rank, bounds,
                     PropagateEngine.alignment(world, p_c, phases, return_chunks, force_grid))


# ------------------------------------------------------------------------------------------------
# autograd wrappers
# ------------------------------------------------------------------------------------------------
class _ShardedProduct(torch.autograd.Function):
    """ys = S xs on the forward operator, d xs = S^T d ys on the transposed one (local rows in, local rows
    out; the exchanges live inside the engine)."""

    @staticmethod
    def forward(ctx, engine, op_fwd, op_bwd, *xs):
        ctx.engine, ctx.op_bwd = engine, op_bwd
        return tuple(engine.run([x.contiguous() for x in xs], op_fwd))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        out = ctx.engine.run([g.contiguous() for g in gs], ctx.op_bwd)
        return (None, None, None) + tuple(out)


class _ShardedMagneticFn(torch.autograd.Function):
    """Forward / backward of one node-sharded MagNetConv / MSConv layer (local rows only): the Chebyshev
    recurrence over engine products, the fused MFMA dense stage on the local rows, dW / db all-reduced."""

    @staticmethod
    def forward(ctx, x_real, x_imag, weight, bias, layer):
        k1 = weight.size(0)
        eng = layer.engine
        ta, tb = [x_real.contiguous()], [x_imag.contiguous()]
        ctx.in_place = None
        if k1 == 2 and layer._reads_in_place(ta[0]):
            # K = 1 (round 5): T_1 stays where the return exchange put it -- the dense stage (and, in the backward pass, its
            # weight-gradient product) reads the receive buffer through a piece layout: no merge pass
            prod = eng.run([ta[0], tb[0]], layer.op_fwd, 1.0, merge=False, input_memo=layer._input_memo,
                           whole_op=layer.op_fwd_whole)
            out_r, out_i = layer._dense_fwd(ta, tb, weight, bias, last_in=prod)
            n_local = layer.plan.n_local
            if n_local < layer.plan.n_pad:
                out_r[n_local:] = 0
                out_i[n_local:] = 0
            ctx.layer, ctx.k1, ctx.has_bias = layer, k1, bias is not None
            ctx.in_place = (prod.off_a, prod.off_b, prod.layout)
            ctx.save_for_backward(weight, ta[0], tb[0], prod.buffer)
            return out_r, out_i
        for k in range(1, k1):
            ya, yb = eng.run([ta[k - 1], tb[k - 1]], layer.op_fwd, 1.0 if k == 1 else 2.0,
                             input_memo=layer._input_memo if k == 1 else None, whole_op=layer.op_fwd_whole if k == 1 else None)
            if k >= 2:                                      # T_k = 2 S T_{k-1} - T_{k-2} on the local rows
                ya, yb = ya - ta[k - 2], yb - tb[k - 2]
            ta.append(ya.contiguous())
            tb.append(yb.contiguous())
        out_r, out_i = layer._dense_fwd(ta, tb, weight, bias)
        n_local = layer.plan.n_local
        if n_local < layer.plan.n_pad:
            # pad rows are isolated nodes that do not exist: their outputs are zero (not the bias), so that a stacked
            # layer sees zero features there again and T_k of a pad row stays zero for every k
            out_r[n_local:] = 0
            out_i[n_local:] = 0
        ctx.layer, ctx.k1, ctx.has_bias = layer, k1, bias is not None
        ctx.save_for_backward(weight, *ta, *tb)
        return out_r, out_i

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_r, g_i):
        layer, k1 = ctx.layer, ctx.k1
        saved = ctx.saved_tensors
        weight = saved[0]
        eng = layer.engine
        last_in = last_out = None
        if ctx.in_place is not None:
            from .dense import PieceOperand
            ta, tb = [saved[1]], [saved[2]]
            last_in = PieceOperand(saved[3], *ctx.in_place)
        else:
            ta, tb = list(saved[1:1 + k1]), list(saved[1 + k1:1 + 2 * k1])
        needs_dx = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        f = ta[0].size(1)
        if k1 == 2 and needs_dx and layer._writes_packed(ta[0]):
            # K = 1 (round 5): dT_1 is consumed by the propagate below and by nothing else -- the dense backward stores it
            # straight into that propagate's send buffers (every replica): no packing pass
            from .dense import PieceOperand
            lay, bufs = eng.send_layout(2, f, ta[0])
            last_out = PieceOperand(bufs[0], 0, lay.slot_floats, lay)
        # (an expanded upstream gradient -- the loss summed the outputs -- stays un-materialised: dense_bwd_raw hands the
        # kernel its one row)
        # upstream gradient on the PAD rows is not part of the graph (the forward zeroes those outputs): the dense
        # backward runs on the real rows only -- dW, db see no pad row whatever the loss or the stacking -- and hands
        # back zero gradient rows for the pad
        da, db, dw, dbias = layer._dense_bwd(ta, tb, weight, g_r, g_i, layer.plan.n_local, last_in=last_in, last_out=last_out)
        # parameter gradients = sum of the per-shard partials: reduced on the communication stream WHILE the backward
        # propagates below run (they do not depend on it); waited for at the end
        pending = [layer.exchange.all_reduce_async(dw)]
        if ctx.has_bias:
            dbias = dbias.contiguous()
            pending.append(layer.exchange.all_reduce_async(dbias))
        gx_r = gx_i = None
        if needs_dx:
            # d T_{k-1} += 2 S^T d T_k ; d T_{k-2} -= d T_k   (k = K .. 2), then gX = d T_0 + S^T d T_1
            fused_add = layer._reads_in_place(ta[0])      # merge + "d T_{k-1} +" in ONE pass over the receive buffer
            for k in range(k1 - 1, 0, -1):
                alpha = 2.0 if k >= 2 else 1.0
                if fused_add:
                    from .dense import gather_pieces
                    if last_out is not None:               # (k = k1 - 1 = 1: the operand is in the send buffers already)
                        prod = eng.run(None, layer.op_bwd, alpha, prepacked=(2, f, ta[0]), merge=False)
                    else:
                        prod = eng.run([da[k], db[k]], layer.op_bwd, alpha, merge=False)
                    da[k - 1], db[k - 1] = gather_pieces(prod, layer.plan.n_pad, f, z=[da[k - 1], db[k - 1]])
                    eng.remark_end()
                else:
                    if last_out is not None:               # (row layout: the all-gather's send buffer was written in place)
                        ra, rb = eng.run(None, layer.op_bwd, alpha, prepacked=(2, f, ta[0]))
                    else:
                        ra, rb = eng.run([da[k], db[k]], layer.op_bwd, alpha)
                    da[k - 1] = da[k - 1] + ra
                    db[k - 1] = db[k - 1] + rb
                if k >= 2:
                    da[k - 2] = da[k - 2] - da[k]
                    db[k - 2] = db[k - 2] - db[k]
            gx_r, gx_i = da[0], db[0]
        for h in pending:
            h.wait()
        return gx_r, gx_i, dw, (dbias if ctx.has_bias else None), None


# ------------------------------------------------------------------------------------------------
# operator builders (HIP)
# ------------------------------------------------------------------------------------------------
def _incident(pid: Tensor, lo: int, hi: int, stride: int, width: int) -> Tensor:
    """Edges with an endpoint among the padded ids {g * stride + [lo, hi) for every g} (width = hi - lo)."""
    a, b = pid[0] % stride, pid[1] % stride
    return ((a >= lo) & (a < hi)) | ((b >= lo) & (b < hi))


def build_magnetic_rows(proto, edge_index: Tensor, edge_weight: Optional[Tensor], plan: ShardPlan, engine, q: float,
                        normalization: Optional[str], lambda_max: float, exchange, how: str = "distributed"):
    """The rows of the scaled magnetic operator this rank multiplies (`engine.block_row_ids`, padded ids), both
    orientations: -> (CSR, (vf_real, vf_imag), (vb_real, vb_imag), global nnz).

    how = "distributed": no rank ever builds the whole operator.  (1) every rank runs the HIP pipeline
    (symmetrise -> sort -> merge -> degree) on the edges incident to its OWNED nodes, which gives the exact
    degree of those nodes; the degrees are all-gathered (N floats).  (2) it runs the pipeline on the edges
    incident to the rows it MULTIPLIES (the same set in the row layout), overrides the degree with the global
    one, evaluates the values and assembles the CSR; only the wanted rows are kept.  The rows are complete, in
    the same (row, col) order and with the same summation order as the single-GPU build.
    how = "global": build the whole operator, slice (single-process rehearsal; cross-check in the tests)."""
    from .utils._laplacian import assemble_operator_csr, laplacian_parts, laplacian_values
    dev = edge_index.device
    pid = plan.to_padded(edge_index)
    kw = proto._laplacian_kwargs()
    rows = engine.block_row_ids(dev)
    w = edge_weight
    if how == "global" or exchange.world_size == 1:
        parts = laplacian_parts(pid, w, plan.n_total, dtype=torch.float32, **kw)
    else:
        own = ((pid[0] >= plan.pad_lo) & (pid[0] < plan.pad_lo + plan.n_pad)) | \
              ((pid[1] >= plan.pad_lo) & (pid[1] < plan.pad_lo + plan.n_pad))
        mine = laplacian_parts(pid[:, own].contiguous(), None if w is None else w[own], plan.n_total,
                               dtype=torch.float32, **kw)
        deg = torch.empty((exchange.world_size, plan.n_pad), dtype=torch.float32, device=dev)
        exchange.all_gather(deg, mine.deg[plan.pad_lo:plan.pad_lo + plan.n_pad].contiguous()).wait()
        if engine.grid:
            lo = engine.i * engine.n_blk
            need = _incident(pid, lo, lo + engine.n_blk, plan.n_pad, engine.n_blk)
            parts = laplacian_parts(pid[:, need].contiguous(), None if w is None else w[need], plan.n_total,
                                    dtype=torch.float32, **kw)
        else:
            parts = mine
        parts.deg = deg.view(-1)
    off_r, off_i, diag, mir_r, mir_i = laplacian_values(parts, q, normalization, mirror=True)
    csr, vf, vb = assemble_operator_csr(parts, off_r, off_i, mir_r, mir_i, diag, float(lambda_max), -1.0)
    sub, vals = take_rows(csr, (vf[0], vf[1], vb[0], vb[1]), rows)
    # entries of the whole operator: every row block counted once (by its column-slice-0 rank)
    local = torch.tensor([float(sub.nnz) if engine.j == 0 else 0.0], dtype=torch.float64, device=dev)
    global_nnz = int(exchange.all_reduce(local).item()) - (plan.n_total - plan.num_nodes)
    return sub, (vals[0], vals[1]), (vals[2], vals[3]), global_nnz


# ------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------
class ShardedMagNetConv(torch.nn.Module):
    """MagNetConv / MSConv (signed=True) over a node-range-sharded graph: each rank owns the rows
    [plan.lo, plan.hi) of the features (inputs, outputs, dense stage).  Parameters are replicated (same seed =>
    same init on every rank); their gradients come back all-reduced.
    forward(x_real_local, x_imag_local) -> local output rows ([n_pad, F]; `shard_rows` / `plan.unshard_rows`).

    layout = "rows" | "grid" | "auto" (grid whenever the width splits into 16-byte column slices and there are
    more than two ranks); phases / return_chunks: the pipeline depth (module docstring; default 2 / 2);
    balance: equal-work node ranges instead of equal-size ones; lambda_max: as the reference's forward argument
    (default 2.0 for 'sym'; normalization=None computes it with eigsh on the whole edge list, like MagNetConv);
    exchange: a DistExchange (default, over `group`) or an EmulatedExchange; build: "distributed" (default: no
    rank assembles the whole operator) or "global" (build everything, keep the rows).
    Reference: nn/directed/MagNetConv.py:122-249, nn/general/MSConv.py:121-230 (the layer); no reference
    counterpart for the sharding."""

    def __init__(self, in_channels: int, out_channels: int, K: int, q: float, num_nodes: int,
                 edge_index: Tensor, edge_weight: Optional[Tensor] = None, normalization: Optional[str] = "sym",
                 bias: bool = True, device=None, group=None, signed: bool = False,
                 absolute_degree: bool = True, layout: str = "auto", grid_cols: Optional[int] = None,
                 phases: Optional[int] = None, return_chunks: Optional[int] = None, balance: bool = True,
                 lambda_max: Optional[float] = None, exchange=None, build: str = "distributed", kernels=None,
                 operator_rows=None, cache_input_exchange: Optional[bool] = None):
        super().__init__()
        from .memo import TensorMemo
        from .nn import MagNetConv, MSConv
        # Opt-in (round 5; PYGSD_SHARD_CACHE_INPUT_EXCHANGE=1): while x_real / x_imag are the same tensors at the same in-place
        # version -- the input features of a first layer, which do not change between training steps -- the forward propagate's
        # INBOUND exchange is not repeated (memo.TensorMemo: weakly held, per version; its opt-outs apply).  Exact, not stale: any
        # write to the features bumps their version.  Off by default: a layer deeper in a model never sees the same tensor twice.
        if cache_input_exchange is None:
            cache_input_exchange = os.environ.get("PYGSD_SHARD_CACHE_INPUT_EXCHANGE", "0") == "1"
        self._input_memo = TensorMemo(2, verify=False) if cache_input_exchange else None      # (opt-in: identity + version)
        self.exchange = exchange if exchange is not None else DistExchange(group)
        self.group = getattr(self.exchange, "group", group)
        world = self.exchange.world_size
        if layout not in ("auto", "rows", "grid"):
            raise ValueError(f"unknown layout {layout!r}")
        p_c = grid_cols if grid_cols is not None else choose_cols(world, in_channels)
        if layout == "auto":
            layout = "grid" if (p_c > 1 and in_channels % (4 * p_c) == 0) else "rows"
        force_grid = layout == "grid" and grid_cols == 1     # the grid SCHEDULE on one column slice (RCCL rehearsal on 1 rank)
        if layout == "rows":
            p_c = 1
        elif (p_c <= 1 and not force_grid) or world % p_c or in_channels % p_c:
            raise ValueError(f"grid of {p_c} column slices does not divide world {world} / width {in_channels}")
        self.layout = layout
        d_ph, d_rc = default_pipeline(world, layout == "grid")
        phases = d_ph if phases is None else phases                 # a count or a tuple of fractions (split_spec)
        return_chunks = d_rc if return_chunks is None else return_chunks
        device = device or edge_index.device
        edge_index = edge_index.to(device)
        edge_weight = None if edge_weight is None else edge_weight.to(device)
        self.plan = make_plan(num_nodes, self.exchange, edge_index, p_c, phases, return_chunks, balance, force_grid)
        self.engine = PropagateEngine(self.plan, self.exchange, p_c, phases, return_chunks, kernels, force_grid)
        proto = (MSConv(in_channels, out_channels, K, q, False, normalization, bias, True, absolute_degree)
                 if signed else MagNetConv(in_channels, out_channels, K, q, False, normalization, True, bias))
        self.weight = proto.weight
        self.bias = proto.bias
        self.in_channels, self.out_channels = in_channels, out_channels
        self.to(device)
        if lambda_max is None:
            if normalization == "sym":
                lambda_max = 2.0
            else:     # the reference computes it from the whole Laplacian (get_magnetic_Laplacian.py:88-92)
                lambda_max = proto._lambda_max_eigsh(edge_index, edge_weight, num_nodes)
        self.lambda_max = float(lambda_max)
        if operator_rows is not None:       # rows prepared by the caller (the CPU tests: no HIP build there)
            csr, vf, vb, self.global_nnz = operator_rows(self.plan, self.engine)
        else:
            if self.exchange.emulated:      # a rehearsal has no peers to gather degrees from
                build = "global"
            csr, vf, vb, self.global_nnz = build_magnetic_rows(proto, edge_index, edge_weight, self.plan, self.engine,
                                                              q, normalization, self.lambda_max, self.exchange, build)
        self.local_nnz = csr.nnz
        self.op_fwd = self.engine.phased(csr, vf, dual=True)
        self.op_bwd = self.engine.phased(csr, vb, dual=True)
        # the forward rows in ONE column block (the arrays the build produced, no copy): what a memoised inbound exchange multiplies
        self.op_fwd_whole = PhasedOperator([(csr, tuple(vf))], True) if self._input_memo is not None else None

    # ---- round 5: the dense kernels address the exchange buffers themselves (no pack / merge passes) ----------------
    def _pieces(self, like: Tensor) -> bool:
        from .dense import dense_supported
        return (self.engine.grid and self.in_channels in (64, 128) and self.out_channels in (64, 128)
                and self.engine.pieces_ok(self.in_channels, like)
                and dense_supported(self.in_channels, self.out_channels, self.weight.size(0)))

    def _reads_in_place(self, like: Tensor) -> bool:
        """Consumers read returned products where the return exchange put them (PYGSD_SHARD_MERGE_ON_READ=0 switches it off)."""
        return _MERGE_ON_READ and self._pieces(like)

    def _writes_packed(self, like: Tensor) -> bool:
        """The dense backward writes dT_K into the send buffers of the propagate that takes it (PYGSD_SHARD_PACKED_BACKWARD=0:
        off).  Also in the row layout: the all-gather's send buffer is a piece layout of one slot."""
        from .dense import dense_supported
        if not _PACKED_BACKWARD:
            return False
        if self.engine.grid:
            return self._pieces(like)
        return (like.is_cuda and like.dtype == torch.float32 and self.in_channels in (64, 128) and self.out_channels in (64, 128)
                and self.engine.phases <= 4 and max(self.engine.phase_rows) * 2 * self.in_channels < (1 << 31)
                and dense_supported(self.in_channels, self.out_channels, self.weight.size(0)))

    @property
    def merge_on_read(self) -> bool:
        return self.weight.size(0) == 2 and self._reads_in_place(self.weight)

    @property
    def packed_backward(self) -> bool:
        return self.weight.size(0) == 2 and self._writes_packed(self.weight)

    # dense stage: the fused MFMA kernels when the shape is tiled by them, library GEMMs otherwise
    def _dense_fwd(self, ta, tb, weight, bias, last_in=None):
        from .dense import dense_fwd_raw, dense_supported
        if ta[0].is_cuda and dense_supported(self.in_channels, self.out_channels, weight.size(0)):
            return dense_fwd_raw(ta, tb, weight, bias, last_in=last_in)
        mm = _dense_mm(ta[0])
        rr = sum(mm(ta[k], weight[k]) for k in range(weight.size(0)))
        ii = sum(mm(tb[k], weight[k]) for k in range(weight.size(0)))
        b = 0 if bias is None else bias
        return rr - ii + b, rr + ii + b

    def _dense_bwd(self, ta, tb, weight, g_r, g_i, rows=None, last_in=None, last_out=None):
        """rows: only the first `rows` rows carry gradient (the rest are pad rows: zero gradient rows come back)."""
        from .dense import dense_bwd_raw, dense_supported
        if ta[0].is_cuda and dense_supported(self.in_channels, self.out_channels, weight.size(0)):
            return dense_bwd_raw(ta, tb, weight, g_r, g_i, rows, last_in=last_in, last_out=last_out)
        n = g_r.size(0)
        rows = n if rows is None else rows
        p, m = (g_r + g_i)[:rows], (g_i - g_r)[:rows]
        k1 = weight.size(0)
        pad = (0, 0, 0, n - rows)
        mm = _dense_mm(p)
        da = [torch.nn.functional.pad(mm(p, weight[k].t()), pad) for k in range(k1)]
        db = [torch.nn.functional.pad(mm(m, weight[k].t()), pad) for k in range(k1)]
        dw = torch.stack([mm(ta[k][:rows].t(), p) + mm(tb[k][:rows].t(), m) for k in range(k1)])
        return da, db, dw, p.sum(0)

    def shard_rows(self, x: Tensor) -> Tensor:
        return self.plan.shard_rows(x)

    def forward(self, x_real_local: Tensor, x_imag_local: Tensor):
        return _ShardedMagneticFn.apply(x_real_local, x_imag_local, self.weight, self.bias, self)


class ShardedOperator:
    """A COO operator out[scatter] += w * x[gather] sharded by node range: this rank keeps the by-target rows (forward) and
    the by-source rows (backward) it multiplies -- its own nodes' in the row layout, its row block's in the grid -- with
    columns = padded ids into the exchanged features.  fp32 or bf16 features (bf16 halves the exchanged bytes as well as the
    gathered ones; with more than one phase its partial products accumulate in fp32 and are rounded once)."""

    def __init__(self, edge_index: Tensor, edge_weight: Optional[Tensor], plan: ShardPlan, engine: PropagateEngine,
                 flow: str = "source_to_target", reduce: str = "add"):
        from .sparse import csr_from_coo, gather_values
        self.plan, self.engine = plan, engine
        g, s = (0, 1) if flow == "source_to_target" else (1, 0)
        pid = plan.to_padded(edge_index)
        gather, scatter = pid[g], pid[s]
        mean = reduce == "mean"
        if mean and engine.phases > 1:
            raise ValueError("reduce='mean' needs an un-phased engine")
        if engine.grid:
            # grid layout (round 4): this rank multiplies ROW BLOCK i -- the i-th 1/p_r of every rank's range, in the
            # engine's product order -- with its column slice of the features; a row keeps all its entries
            rows = engine.block_row_ids(pid.device)
            where = torch.full((plan.n_total,), -1, dtype=torch.long, device=pid.device)
            where[rows] = torch.arange(rows.numel(), device=pid.device)
            at_t, at_s = where[scatter], where[gather]
            keep_t, keep_s = (at_t >= 0).nonzero(as_tuple=True)[0], (at_s >= 0).nonzero(as_tuple=True)[0]
            fwd = csr_from_coo(at_t[keep_t], gather[keep_t], engine.block_rows, plan.n_total)
            bwd = csr_from_coo(at_s[keep_s], scatter[keep_s], engine.block_rows, plan.n_total)
        else:
            lo, hi = plan.pad_lo, plan.pad_lo + plan.n_pad
            keep_t = ((scatter >= lo) & (scatter < hi)).nonzero(as_tuple=True)[0]
            fwd = csr_from_coo(scatter[keep_t] - lo, gather[keep_t], plan.n_pad, plan.n_total)
            keep_s = ((gather >= lo) & (gather < hi)).nonzero(as_tuple=True)[0]
            bwd = csr_from_coo(gather[keep_s] - lo, scatter[keep_s], plan.n_pad, plan.n_total)
        w = edge_weight
        if mean:                                                       # backward of mean: 1 / in-degree per entry
            ones = torch.ones(pid.size(1), dtype=torch.float32, device=pid.device)
            deg = torch.zeros(plan.n_total, dtype=torch.float32, device=pid.device).index_add_(0, scatter, ones)
            inv = (1.0 / deg.clamp(min=1))[scatter]
            wb = inv if w is None else w.float() * inv
        else:
            wb = w
        vf = None if w is None else gather_values(w[keep_t], fwd.perm)
        vb = None if wb is None else gather_values(wb[keep_s], bwd.perm)
        self.local_nnz = int(keep_t.numel())
        self.op_fwd = PhasedOperator(split_phases(fwd, _vals(vf, fwd), plan.n_pad, engine.phases, plan.world_size,
                                                  engine.phase_bounds), False, mean)
        self.op_bwd = PhasedOperator(split_phases(bwd, _vals(vb, bwd), plan.n_pad, engine.phases, plan.world_size,
                                                  engine.phase_bounds), False)


def _vals(v: Optional[Tensor], csr: CSR) -> Tuple[Tensor]:
    if v is None:
        v = torch.ones(csr.nnz, dtype=torch.float32, device=csr.col.device)
    return (v,)


class _GradSync:
    """Parameter gradients of a layer with replicated parameters, summed over the ranks in an order that does NOT
    depend on autograd.  Per-parameter all-reduce hooks fire in the order autograd makes the gradients ready; that
    order is the same on every rank only if the ranks run the same graph under the same scheduling, and when it is not,
    two ranks pair up DIFFERENT parameters in one all-reduce (seen at the C5 size under load: one rank's conv2.weight
    share summed into conv1.weight, a 12 % error).  Here the hooks only remember each incoming local gradient; a
    callback that autograd runs at the END of the backward pass all-reduces them in parameter order and corrects
    `.grad` by (sum over ranks - local share) -- correct under gradient accumulation too, since only the share of THIS
    backward is exchanged."""

    def _install_grad_sync(self):
        self._sync_params = list(self.parameters())
        self._sync_local = {}
        self._sync_seen = {}           # k -> (the .grad object, its version) found when this pass's first share arrived
        self._sync_task = None         # autograd graph task the pending shares belong to
        for k, prm in enumerate(self._sync_params):
            prm.register_hook(lambda grad, k=k: self._remember(grad, k))

    def _begin_pass(self):
        task = torch._C._current_graph_task_id()
        if task != self._sync_task:
            # a new backward pass.  Shares left over from a pass that aborted after its first hook (OOM, an exception
            # in a later node: the engine never ran that pass's callback) are dropped here instead of blocking every
            # later exchange; the callback is queued once per graph task, whatever state the last one ended in.
            self._sync_task, self._sync_local, self._sync_seen = task, {}, {}
            torch.autograd.Variable._execution_engine.queue_callback(self._exchange_gradients)

    def _arm(self, out):
        """Called on what forward() returns.  The end-of-pass exchange is a COLLECTIVE: every rank has to run it in every
        backward pass that reaches this layer -- also a rank none of whose parameters receives a gradient in that pass (its
        peers would wait in the all-reduces for ever).  A hook on the layer's outputs queues the callback as soon as the pass
        reaches the layer, whatever happens to the parameters afterwards."""
        def hook(grad):
            self._begin_pass()
            return grad
        for t in (out if isinstance(out, tuple) else (out,)):
            if t.requires_grad:
                t.register_hook(hook)
        return out

    def _remember(self, grad, k):
        self._begin_pass()
        if k not in self._sync_seen:
            held = self._sync_params[k].grad
            self._sync_seen[k] = (held, None if held is None else held._version)
        prev = self._sync_local.get(k)
        self._sync_local[k] = grad if prev is None else prev + grad     # a parameter used twice in one graph
        return grad

    def _exchange_gradients(self):
        local, seen = self._sync_local, self._sync_seen
        self._sync_local, self._sync_seen, self._sync_task = {}, {}, None
        if not self._sync_params:
            return
        with torch.no_grad():
            # which parameters received a share on ANY rank: one small all-reduce and the pass's one host read.  A parameter no
            # rank used keeps the `.grad` it had -- None under zero_grad(set_to_none=True), exactly as in the single-device
            # reference, so that an optimiser skips it (an explicit zero gradient would still be decayed / given momentum).
            flags = torch.tensor([1.0 if k in local else 0.0 for k in range(len(self._sync_params))], dtype=torch.float32,
                                 device=self._sync_params[0].device)
            used = self.exchange.all_reduce(flags).tolist()
            stale = None
            for k, prm in enumerate(self._sync_params):                  # fixed order on every rank
                if used[k] == 0:
                    continue
                mine = local.get(k)
                if mine is None:                       # other ranks' shares of a parameter this rank's graph skipped
                    total = self.exchange.all_reduce(torch.zeros_like(prm))
                    # with `.grad` still None the total BECOMES the gradient: every rank has to end the pass with the same
                    # `.grad`, or the replicated parameters drift apart at the next optimiser step
                    if prm.grad is not None:
                        prm.grad.add_(total)
                    elif prm.requires_grad:
                        prm.grad = total
                    continue
                total = self.exchange.all_reduce(mine.detach().clone().contiguous())
                held, version = seen[k]
                if prm.grad is None or (prm.grad is held and prm.grad._version == version):
                    stale = k                          # nothing was accumulated into .grad: torch.autograd.grad(...)
                    continue
                prm.grad.add_(total - mine)
            if stale is not None:
                # every rank reaches this after the same collectives, so raising cannot leave a peer inside one
                raise RuntimeError(
                    "sharded layers all-reduce parameter gradients into `.grad` at the end of `.backward()`; "
                    "torch.autograd.grad(...) would hand back this rank's share only -- use .backward() "
                    f"(parameter #{stale} received a gradient that was not accumulated)")


class ShardedDiGCNConv(_GradSync, torch.nn.Module):
    """DiGCNConv (out = S^T (x W) + b, reference nn/directed/DiGCNConv.py:54-94) over a node-range-sharded
    graph; fp32 or bf16 (`.to(torch.bfloat16)`: BASELINE config "DiGCN_Inception_Block ... bf16, 8xMI355X").
    Parameters are replicated; their gradients are all-reduced at the end of every backward pass (`_GradSync`)."""

    def __init__(self, in_channels: int, out_channels: int, num_nodes: int, edge_index: Tensor,
                 edge_weight: Tensor, bias: bool = True, device=None, group=None, exchange=None,
                 phases: Optional[int] = None, balance: bool = True, plan: Optional[ShardPlan] = None, kernels=None,
                 grid_cols: int = 1, return_chunks: int = 1):
        super().__init__()
        from .nn import DiGCNConv
        proto = DiGCNConv(in_channels, out_channels, bias=bias)
        self.weight, self.bias = proto.weight, proto.bias
        self.in_channels, self.out_channels = in_channels, out_channels
        device = device or edge_index.device
        self.to(device)
        if edge_weight is None:
            raise RuntimeError('Normalized adj matrix cannot be None. Please obtain the adj matrix in preprocessing.')
        self.exchange = exchange if exchange is not None else DistExchange(group)
        edge_index = edge_index.to(device)
        phases = 1 if phases is None else int(phases)       # measured: a second phase does not pay for one-operand rows
        # grid_cols > 1 (round 4): p_r x p_c process grid -- column-slice all-to-all in, products home over all links
        self.plan = plan or make_plan(num_nodes, self.exchange, edge_index, grid_cols, phases, return_chunks, balance)
        self.engine = PropagateEngine(self.plan, self.exchange, grid_cols, phases, return_chunks, kernels)
        self.op = ShardedOperator(edge_index, edge_weight.to(device), self.plan, self.engine)
        self._install_grad_sync()

    def shard_rows(self, x: Tensor) -> Tensor:
        return self.plan.shard_rows(x)

    def aggregate(self, xw_local: Tensor) -> Tensor:
        (out,) = _ShardedProduct.apply(self.engine, [self.op.op_fwd], [self.op.op_bwd], xw_local)
        return out

    def forward(self, x_local: Tensor) -> Tensor:
        from .dense import tall_linear
        out = self.aggregate(tall_linear(x_local, self.weight))
        out = out if self.bias is None else out + self.bias
        return self._arm(_zero_pad_rows(self.plan, out))


def _dense_mm(like: Tensor):
    """The product routine of the sharded layers' dense stage for shapes the MFMA kernels do not tile: the generic HIP GEMM
    on the device, torch.matmul for the CPU restatements the gloo tests run."""
    if like.is_cuda and like.dtype == torch.float32:
        from .dense import gemm
        return lambda a, b: gemm(a.detach(), b.detach())
    return torch.matmul


def _zero_pad_rows(plan: ShardPlan, t: Tensor) -> Tensor:
    """Pad rows -> 0.  Applied on EVERY rank, also the one whose range has no pad rows, so that the ranks run the same
    autograd graph (the parameter all-reduce no longer depends on that -- `_GradSync` -- but identical graphs keep the
    ranks' kernel sequences, and with them their collectives, in step)."""
    mask = torch.zeros((plan.n_pad, 1), dtype=t.dtype, device=t.device)
    mask[:plan.n_local] = 1
    return t * mask


class ShardedDiGCNInceptionBlock(_GradSync, torch.nn.Module):
    """DiGCN_InceptionBlock (reference nn/directed/DiGCN_Inception_Block.py:9-47: x0 = Linear(x), x1 / x2 =
    DiGCNConv on the first- / second-order proximity operators) over a node-range-sharded graph, fp32 or bf16.
    The two convolutions share ONE exchange per propagate: the projections x W1 and x W2 are packed side by side
    into one all-gather, and each operator reads its own column half of the gathered rows.
    forward(x_local) -> (x0, x1, x2) local rows.  Same state_dict keys as the reference block
    (ln.weight, ln.bias, conv1.weight, conv1.bias, conv2.weight, conv2.bias)."""

    def __init__(self, in_dim: int, out_dim: int, num_nodes: int, edge_index: Tensor, edge_weight: Tensor,
                 edge_index2: Tensor, edge_weight2: Tensor, device=None, group=None, exchange=None,
                 phases: Optional[int] = None, balance: bool = True, kernels=None, grid_cols: int = 1,
                 return_chunks: int = 1):
        super().__init__()
        from .nn import DiGCNConv
        self.ln = torch.nn.Linear(in_dim, out_dim)
        self.conv1, self.conv2 = DiGCNConv(in_dim, out_dim), DiGCNConv(in_dim, out_dim)
        device = device or edge_index.device
        self.to(device)
        self.exchange = exchange if exchange is not None else DistExchange(group)
        edge_index, edge_index2 = edge_index.to(device), edge_index2.to(device)
        both = torch.cat([edge_index, edge_index2], dim=1)
        phases = 1 if phases is None else int(phases)       # see ShardedDiGCNConv
        self.plan = make_plan(num_nodes, self.exchange, both, grid_cols, phases, return_chunks, balance)
        self.engine = PropagateEngine(self.plan, self.exchange, grid_cols, phases, return_chunks, kernels)
        self.op1 = ShardedOperator(edge_index, edge_weight.to(device), self.plan, self.engine)
        self.op2 = ShardedOperator(edge_index2, edge_weight2.to(device), self.plan, self.engine)
        self._install_grad_sync()

    def shard_rows(self, x: Tensor) -> Tensor:
        return self.plan.shard_rows(x)

    def forward(self, x_local: Tensor):
        from .dense import tall_linear
        x0 = tall_linear(x_local, self.ln.weight.t(), self.ln.bias)
        p1, p2 = tall_linear(x_local, self.conv1.weight), tall_linear(x_local, self.conv2.weight)
        x1, x2 = _ShardedProduct.apply(self.engine, [self.op1.op_fwd, self.op2.op_fwd],
                                       [self.op1.op_bwd, self.op2.op_bwd], p1, p2)
        if self.conv1.bias is not None:
            x1, x2 = x1 + self.conv1.bias, x2 + self.conv2.bias
        return self._arm(tuple(_zero_pad_rows(self.plan, t) for t in (x0, x1, x2)))


class ShardedSGCNConv(_GradSync, torch.nn.Module):
    """SGCNConv (reference nn/signed/SGCNConv.py:94-126: mean over positive / negative incoming edges, concatenated with
    the node's own features, Linear) over a node-range-sharded graph, for in_dim >= out_dim (the narrowing layers of SGCN).
    As the un-sharded layer it multiplies first: one local product x [own_b | own_u | agg_1 | ...] (the Linear's blocks as
    column blocks), then every aggregated block -- out_dim wide -- goes through ONE shared all-gather and its own
    mean-reducing operator rows; the exchange moves m * out_dim columns per node (m = 2 first / 4 deep aggregation) instead
    of the m input blocks.  Parameters (`lin_b`, `lin_u`: the reference's state_dict keys) are replicated, their gradients
    all-reduced at the end of every backward pass."""

    def __init__(self, in_dim: int, out_dim: int, first_aggr: bool, num_nodes: int, pos_edge_index: Tensor,
                 neg_edge_index: Tensor, bias: bool = True, device=None, group=None, exchange=None, balance: bool = True,
                 kernels=None):
        super().__init__()
        if in_dim < out_dim:
            raise NotImplementedError("ShardedSGCNConv multiplies before it aggregates: in_dim >= out_dim")
        self.in_dim, self.out_dim, self.first_aggr = in_dim, out_dim, first_aggr
        k = 2 if first_aggr else 3
        self.lin_b = torch.nn.Linear(k * in_dim, out_dim, bias)
        self.lin_u = torch.nn.Linear(k * in_dim, out_dim, bias)
        device = device or pos_edge_index.device
        self.to(device)
        self.exchange = exchange if exchange is not None else DistExchange(group)
        pos, neg = pos_edge_index.to(device), neg_edge_index.to(device)
        self.plan = make_plan(num_nodes, self.exchange, torch.cat([pos, neg], dim=1), 1, 1, 1, balance)
        self.engine = PropagateEngine(self.plan, self.exchange, 1, 1, 1, kernels)      # mean: one phase
        self.op_pos = ShardedOperator(pos, None, self.plan, self.engine, reduce="mean")
        self.op_neg = ShardedOperator(neg, None, self.plan, self.engine, reduce="mean")
        self._install_grad_sync()

    def shard_rows(self, x: Tensor) -> Tensor:
        return self.plan.shard_rows(x)

    def forward(self, x_local: Tensor) -> Tensor:
        from .dense import tall_linear
        f, o = self.in_dim, self.out_dim
        wb, wu = self.lin_b.weight, self.lin_u.weight
        if self.first_aggr:             # columns: own_b | own_u | agg_b(pos) | agg_u(neg)
            w_big = torch.cat([wb[:, f:].t(), wu[:, f:].t(), wb[:, :f].t(), wu[:, :f].t()], dim=1)
            spec = ((self.op_pos, 0), (self.op_neg, 1))
        else:                           # x = [lo | hi]; columns: own_b | own_u | pos_b | neg_b | pos_u | neg_u
            zeros = wb.new_zeros(f, o)
            top = torch.cat([wb[:, 2 * f:].t(), zeros, wb[:, :f].t(), zeros, zeros, wu[:, f:2 * f].t()], dim=1)
            bot = torch.cat([zeros, wu[:, 2 * f:].t(), zeros, wb[:, f:2 * f].t(), wu[:, :f].t(), zeros], dim=1)
            w_big = torch.cat([top, bot], dim=0)
            spec = ((self.op_pos, 0), (self.op_neg, 0), (self.op_pos, 1), (self.op_neg, 1))
        bias = None
        if self.lin_b.bias is not None:
            bias = torch.cat([self.lin_b.bias, self.lin_u.bias, self.lin_b.bias.new_zeros(len(spec) * o)])
        y = tall_linear(x_local, w_big, bias)
        parts = [y[:, (2 + j) * o:(3 + j) * o] for j in range(len(spec))]
        agg = _ShardedProduct.apply(self.engine, [op.op_fwd for op, _ in spec], [op.op_bwd for op, _ in spec], *parts)
        halves = [y[:, :o], y[:, o:2 * o]]
        for (_, half), a in zip(spec, agg):
            halves[half] = halves[half] + a
        return self._arm(_zero_pad_rows(self.plan, torch.cat(halves, dim=1)))


class _HopTerm(torch.autograd.Function):
    """w * x for a one-element hop weight w: the weight's gradient <g, x> -- a sum over all local rows and columns -- is
    taken in float64 (pygsd_dots_f32, as the single-device SIMPA does; an fp32 tree leaves ~1e-5 on sums whose partial
    sums are in the hundreds: tests/test_gpu_sharded.py -k fuzz), the per-rank sums are then added by `_GradSync`."""

    @staticmethod
    def forward(ctx, w, x):
        ctx.save_for_backward(w, x)
        return x * w

    @staticmethod
    def backward(ctx, g):
        w, x = ctx.saved_tensors
        gw = gx = None
        if ctx.needs_input_grad[1]:
            gx = g * w
        if ctx.needs_input_grad[0]:
            if g.is_cuda and g.dim() == 2 and g.size(1) % 4 == 0 and g.dtype == torch.float32 and x.dtype == torch.float32:
                from .nn.signed.SIMPA import dots
                gw = dots(g, [x]).reshape(w.shape)
            else:
                gw = (g.double() * x.double()).sum().to(w.dtype).reshape(w.shape)
        return gw, gx


class _HopWeights:
    """wp[h] * x through `_HopTerm`."""

    def __init__(self, w: Tensor):
        self.w = w

    def __getitem__(self, h):
        return _Hop(self.w[h])


class _Hop:
    def __init__(self, w: Tensor):
        self.w = w

    def __mul__(self, x: Tensor) -> Tensor:
        return _HopTerm.apply(self.w, x)


class ShardedSIMPA(_GradSync, torch.nn.Module):
    """SIMPA (reference nn/signed/SIMPA.py:52-144, the aggregation of SSSNET) over a node-range-sharded graph: the hop
    schedule of the un-sharded layer, every product of the random-walk operators A_p = D^-1 (A+ + fill I), A_n = D^-1 A-
    (nn/general/conv_base.py:12-31, aggregation at edge_index[0]) an all-gather + this rank's operator rows.  The two
    operators are normalised on every rank from the whole edge list (`normalise`, default utils._norm.conv_norm_rw: a
    row's degree needs only the row's own entries, so sharding that build is possible, not done).  Hop weights
    (`_w_p`, `_w_n` / `_w_sp` ... `_w_tn`: the reference's state_dict keys) are replicated, their gradients all-reduced."""

    def __init__(self, hop: int, fill_value: float, num_nodes: int, edge_index_p: Tensor, edge_weight_p: Optional[Tensor],
                 edge_index_n: Tensor, edge_weight_n: Optional[Tensor], directed: bool = False, device=None, group=None,
                 exchange=None, balance: bool = True, kernels=None, normalise=None):
        super().__init__()
        from .nn.general.conv_base import flipped_edge_index
        if normalise is None:
            from .utils._norm import conv_norm_rw as normalise
        self._hop_p, self._hop_n = hop + 1, int((1 + hop) * hop / 2)
        self._undirected = not directed
        names = ("_w_p", "_w_n") if self._undirected else ("_w_sp", "_w_sn", "_w_tp", "_w_tn")
        for name in names:
            rows = self._hop_n if name.endswith("n") else self._hop_p
            self.register_parameter(name, torch.nn.Parameter(torch.ones(rows, 1)))
        device = device or edge_index_p.device
        self.to(device)
        self.exchange = exchange if exchange is not None else DistExchange(group)
        ei_p, ei_n = edge_index_p.to(device), edge_index_n.to(device)
        w_p = None if edge_weight_p is None else edge_weight_p.to(device)
        w_n = None if edge_weight_n is None else edge_weight_n.to(device)
        self.plan = make_plan(num_nodes, self.exchange, torch.cat([ei_p, ei_n], dim=1), 1, 1, 1, balance)
        self.engine = PropagateEngine(self.plan, self.exchange, 1, 1, 1, kernels)

        def operator(ei, w, fill):
            nei, nw = normalise(ei, fill, w, num_nodes)
            return ShardedOperator(nei, nw, self.plan, self.engine, flow="target_to_source")

        self.ops = {"p": operator(ei_p, w_p, fill_value), "n": operator(ei_n, w_n, 0.0)}
        if directed:
            self.ops["tp"] = operator(flipped_edge_index(ei_p), w_p, fill_value)
            self.ops["tn"] = operator(flipped_edge_index(ei_n), w_n, 0.0)
        self._install_grad_sync()

    def shard_rows(self, x: Tensor) -> Tensor:
        return self.plan.shard_rows(x)

    def _product(self, key: str, x: Tensor) -> Tensor:
        op = self.ops[key]
        (y,) = _ShardedProduct.apply(self.engine, [op.op_fwd], [op.op_bwd], x)
        return y

    def _stream(self, kp: str, kn: str, x_pos: Tensor, x_neg: Tensor, wp: Tensor, wn: Tensor) -> Tensor:
        """[feat_p | feat_n] of one (positive, negative) pair in the reference's accumulation order (SIMPA.py:77-93)."""
        wp, wn = _HopWeights(wp), _HopWeights(wn)
        feat_p, feat_n = wp[0] * x_pos, None
        cur_p, aux_n, j, last = x_pos, x_neg, 0, self._hop_p - 1
        for h in range(self._hop_p):
            if h > 0:
                cur_p = self._product(kp, cur_p)
                if h != last:          # the reference also advances aux_n at the last hop, but never reads it again
                    aux_n = self._product(kp, aux_n)
                feat_p = feat_p + wp[h] * cur_p
            if h != last:
                cur_n = self._product(kn, aux_n)
                feat_n = wn[j] * cur_n if feat_n is None else feat_n + wn[j] * cur_n
                j += 1
                for _ in range(self._hop_p - 2 - h):
                    cur_n = self._product(kp, cur_n)
                    feat_n = feat_n + wn[j] * cur_n
                    j += 1
        return torch.cat([feat_p, torch.zeros_like(feat_p) if feat_n is None else feat_n], dim=1)

    def forward(self, x_p: Tensor, x_n: Tensor, x_pt: Optional[Tensor] = None, x_nt: Optional[Tensor] = None) -> Tensor:
        """Local rows in (x_p, x_n[, x_pt, x_nt]: [n_pad, F]), local rows of [feat_p | feat_n (| target streams)] out."""
        if self._undirected:
            return self._arm(_zero_pad_rows(self.plan, self._stream("p", "n", x_p, x_n, self._w_p, self._w_n)))
        source = self._stream("p", "n", x_p, x_n, self._w_sp, self._w_sn)
        target = self._stream("tp", "tn", x_pt, x_nt, self._w_tp, self._w_tn)
        return self._arm(_zero_pad_rows(self.plan, torch.cat([source, target], dim=1)))


# ------------------------------------------------------------------------------------------------
# helpers kept for callers / tests
# ------------------------------------------------------------------------------------------------
def all_gather_rows(x_local: Tensor, group=None) -> Tensor:
    """[n_pad, C] per rank -> [world * n_pad, C], rank-major (padded-id order under ShardPlan)."""
    ex = DistExchange(group)
    x_local = x_local.contiguous()
    out = x_local.new_empty((ex.world_size,) + tuple(x_local.shape))
    ex.all_gather(out, x_local).wait()
    return out.view((ex.world_size * x_local.size(0),) + tuple(x_local.shape[1:]))
