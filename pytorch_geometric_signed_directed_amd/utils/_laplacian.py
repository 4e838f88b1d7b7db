"""Device-side build of the (signed) magnetic Laplacian operator -- SURVEY.md 8(a) rows a3/a4 --
through the HIP pipeline of csrc/laplacian.hip (include/pygsd_hip.h: pygsd_maglap_*).

Follows utils/directed/get_magnetic_Laplacian.py:10-93 and
utils/general/get_magnetic_signed_Laplacian.py:10-98 of the reference: drop self loops ->
symmetrise -> coalesce(add) of [w, +-w(, |w|)] -> A_s = sum/2, Theta_arg = w_uv - w_vu -> degree ->
D^-1/2 A_s D^-1/2 (.) exp(i 2 pi q Theta_arg) -> L = I - H (sym) or D - A_s (.) exp(...) (None).

One host round-trip is inherent: the number of distinct symmetrised entries sizes the outputs
(the reference pays the same for coalesce's boolean mask).
"""
import ctypes
import math
import os
import threading
from typing import Optional

import torch

from .. import _cabi
from .._cabi import check, ptr, stream_ptr
from ..memo import TensorMemo

Tensor = torch.Tensor


class LaplacianParts:
    """Symmetrised pattern + the ingredients of the operator values (all COO, sorted by (row, col))."""
    __slots__ = ("index", "a_sym", "theta", "deg", "n", "off_ptr", "differentiable")

    def __init__(self, index, a_sym, theta, deg, n, off_ptr, differentiable=False):
        self.index, self.a_sym, self.theta, self.deg, self.n = index, a_sym, theta, deg, n
        self.off_ptr = off_ptr          # int32 [n + 1]: first sorted entry of each row
        self.differentiable = differentiable    # a_sym / theta / deg carry the autograd graph of edge_weight


def laplacian_parts(edge_index: Tensor, edge_weight: Optional[Tensor], n: int, signed: bool,
                    absolute_degree: bool, dtype=None) -> LaplacianParts:
    _cabi.require_gpu(edge_index, edge_weight)
    if edge_index.dtype != torch.int64:
        raise TypeError("edge_index must be int64 (torch.long)")
    dev = edge_index.device
    row, col = edge_index[0].contiguous(), edge_index[1].contiguous()
    e = row.numel()
    # the sort keys are row * n + col and the later stages index deg[] / off_ptr[] by these ids: an id outside
    # [0, n) (wrong num_nodes, edge_index of a larger graph) must raise as the reference's scatter does
    _cabi.check_node_ids((n, row), (n, col))
    w = None
    if edge_weight is not None:
        w = edge_weight.detach().reshape(-1).contiguous()
        if w.dtype != torch.float32:
            w = w.float()
        if w.numel() != e:
            raise ValueError(f"edge_weight has {w.numel()} entries for {e} edges")
    lib = _cabi.lib()
    with _cabi.on_device(dev):
        need = ctypes.c_size_t(0)
        check(lib.pygsd_maglap_workspace(e, ctypes.byref(need)), "pygsd_maglap_workspace")
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        check(lib.pygsd_maglap_sort(ptr(row), ptr(col), e, n, ptr(ws), need.value, ptr(count), stream_ptr()),
              "pygsd_maglap_sort")
        es = int(count.item())                      # the one host round-trip
        index = torch.empty((2, es), dtype=torch.int64, device=dev)
        a_sym = torch.empty(es, dtype=torch.float32, device=dev)
        theta = torch.empty(es, dtype=torch.float32, device=dev)
        deg = torch.empty(n, dtype=torch.float32, device=dev)
        off_ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        check(lib.pygsd_maglap_merge(ptr(w), e, n, 1 if signed else 0, 1 if absolute_degree else 0, es, ptr(ws),
                                     need.value, ptr(index[0]) if es else None, ptr(index[1]) if es else None,
                                     ptr(a_sym), ptr(theta), ptr(deg), ptr(off_ptr), stream_ptr()),
              "pygsd_maglap_merge")
    return LaplacianParts(index, a_sym, theta, deg, n, off_ptr)


def differentiable_parts(parts: LaplacianParts, edge_index: Tensor, edge_weight: Tensor, signed: bool,
                          absolute_degree: bool) -> LaplacianParts:
    """The same sorted pattern as `parts` (built by the HIP pipeline from the detached weights) with a_sym,
    theta_arg and the degree recomputed by differentiable tensor ops, for an `edge_weight` that requires grad:
    the reference's get_magnetic_(signed_)Laplacian is differentiable w.r.t. edge_weight (coalesce / scatter_add
    of [w, +-w(, |w|)], get_magnetic_Laplacian.py:52-66, get_magnetic_signed_Laplacian.py:52-73).  Every listed
    non-loop edge u -> v adds w/2 to A_s at (u, v) and (v, u) and +w / -w to the phase argument there; the slot of
    an entry in the sorted pattern is a binary search over its (row * n + col) keys."""
    n, es = parts.n, parts.a_sym.numel()
    row, col = parts.index[0], parts.index[1]
    keys = row * n + col                                            # ascending (sorted by (row, col))
    u, v = edge_index[0], edge_index[1]
    keep = u != v
    u, v = u[keep], v[keep]
    w = edge_weight.reshape(-1)[keep].to(torch.float32)
    s_uv = torch.searchsorted(keys, u * n + v)
    s_vu = torch.searchsorted(keys, v * n + u)
    zeros = torch.zeros(es, dtype=torch.float32, device=w.device)
    a_sym = (zeros.index_add(0, s_uv, w).index_add(0, s_vu, w)) * 0.5
    theta = zeros.index_add(0, s_uv, w).index_add(0, s_vu, -w)
    if not signed:
        per_entry = a_sym
    elif absolute_degree:
        per_entry = (zeros.index_add(0, s_uv, w.abs()).index_add(0, s_vu, w.abs())) * 0.5
    else:
        per_entry = a_sym.abs()
    deg = torch.zeros(n, dtype=torch.float32, device=w.device).index_add(0, row, per_entry)
    return LaplacianParts(parts.index, a_sym, theta, deg, n, parts.off_ptr, differentiable=True)


def assemble_operator_csr(parts: LaplacianParts, off_real: Tensor, off_imag: Tensor, mir_real: Tensor,
                          mir_imag: Tensor, diag: Tensor, lambda_max: float, diag_shift: float = -1.0):
    """Compute layout of the scaled operator 2 L / lambda_max + diag_shift I on the symmetrised pattern:
    one shared int32 CSR (diagonal merged in, columns ascending) + the values for the by-target (forward)
    and by-source (backward) products.  Returns (CSR, (vf_real, vf_imag), (vb_real, vb_imag))."""
    from ..sparse import CSR
    n, es = parts.n, parts.a_sym.numel()
    dev = diag.device
    nnz = es + n
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    col = torch.empty(nnz, dtype=torch.int32, device=dev)
    vals = torch.empty((4, max(nnz, 1)), dtype=torch.float32, device=dev)
    with _cabi.on_device(dev):
        check(_cabi.lib().pygsd_maglap_assemble_csr(ptr(parts.index[0]) if es else None,
                                                    ptr(parts.index[1]) if es else None, ptr(off_real), ptr(off_imag),
                                                    ptr(mir_real), ptr(mir_imag), ptr(diag), ptr(parts.off_ptr), es, n,
                                                    float(lambda_max), float(diag_shift), ptr(rowptr), ptr(col),
                                                    ptr(vals[0]), ptr(vals[1]), ptr(vals[2]), ptr(vals[3]),
                                                    stream_ptr()), "pygsd_maglap_assemble_csr")
    csr = CSR(n, n, nnz, rowptr, col, None)
    return csr, (vals[2, :nnz], vals[3, :nnz]), (vals[0, :nnz], vals[1, :nnz])


def laplacian_values(parts: LaplacianParts, q, normalization: Optional[str], mirror: bool = False):
    """(off_real, off_imag, diag) of L [+ (mir_real, mir_imag): the values of the mirrored entries].
    A float q runs the HIP kernel; a tensor q that requires grad (trainable_q), or ingredients that carry the
    gradient of edge_weight (`differentiable_parts`), are evaluated with differentiable element-wise tensor ops
    (no mirror values there)."""
    sym = normalization is not None
    diag = torch.ones_like(parts.deg) if sym else parts.deg
    if (isinstance(q, torch.Tensor) and q.requires_grad) or parts.differentiable:
        if mirror:
            raise ValueError("mirror values are only produced for a fixed q and weights without gradient")
        row, col = parts.index[0], parts.index[1]
        phase_arg = (2 * math.pi * q) * parts.theta
        if sym:
            dis = parts.deg.pow(-0.5)
            dis = dis.masked_fill(dis == float("inf"), 0)
            mag = dis[row] * parts.a_sym * dis[col]
        else:
            mag = parts.a_sym
        return -(mag * torch.cos(phase_arg)), -(mag * torch.sin(phase_arg)), diag
    qf = float(q.detach().item()) if isinstance(q, torch.Tensor) else float(q)
    es = parts.a_sym.numel()
    off_r = torch.empty_like(parts.a_sym)
    off_i = torch.empty_like(parts.a_sym)
    mir_r = torch.empty_like(parts.a_sym) if mirror else None
    mir_i = torch.empty_like(parts.a_sym) if mirror else None
    if es:
        with _cabi.on_device(parts.a_sym.device):
            check(_cabi.lib().pygsd_maglap_values(ptr(parts.index[0]), ptr(parts.index[1]), ptr(parts.a_sym),
                                                  ptr(parts.theta), ptr(parts.deg), es, qf, 1 if sym else 0,
                                                  ptr(off_r), ptr(off_i), ptr(mir_r), ptr(mir_i), stream_ptr()),
                  "pygsd_maglap_values")
    if mirror:
        return off_r, off_i, diag, mir_r, mir_i
    return off_r, off_i, diag


_PINNED = threading.local()



def _pinned_info(dev):
    """(pinned int64[4], event `ready`, side stream, event `mid`) for the device -> host read of the fused build; one per
    (thread, device) -- thread-local, so the buffers of a pool's worker threads die with their threads."""
    table = getattr(_PINNED, "table", None)
    if table is None:
        table = _PINNED.table = {}
    hit = table.get(dev.index)
    if hit is None:
        hit = table[dev.index] = (torch.empty(4, dtype=torch.int64, pin_memory=True), torch.cuda.Event(),
                                  torch.cuda.Stream(device=dev), torch.cuda.Event())
    return hit


def _queue_info_read(info: Tensor, dev):
    """Queue the device -> host copy of a build's `info` words BEHIND what the current stream holds now, on a side stream -- the
    kernels the caller launches next (the rest of the build) do not wait for the copy, and the host does not wait for them:
    `ready.synchronize()` returns as soon as the words have landed.  -> (pinned int64[4], ready)."""
    host_info, ready, side, mid = _pinned_info(dev)
    mid.record()
    side.wait_event(mid)
    with torch.cuda.stream(side):
        host_info.copy_(info, non_blocking=True)
        ready.record()
    return host_info, ready


_UNIT_BUILD = os.environ.get("PYGSD_TWO_STAGE_BUILD", "0") != "1"


_SIGNED_UNIT_BUILD = os.environ.get("PYGSD_SIGNED_UNIT_BUILD", "1") != "0"
_NOT_PM1 = TensorMemo(8, verify=False)       # (which build to try first: never a wrong result) weight tensors the +-1 build has turned down (weakly held, per in-place version)
_NOT_BUCKETS = TensorMemo(8, verify=False)   # (edge_index, edge_weight) pairs the weighted bucket form has turned down (duplicates, hub rows)
# A call site (a layer) whose builds the +-1 form has turned down this many times in a row -- fresh real-valued weight tensors
# on every uncached forward, which the per-tensor memo above cannot recognise -- stops offering them: each refusal costs a
# counting pass, a scan and a host round trip in front of the build that then runs.
_PM1_SITE_REFUSALS = 2


def set_signed_unit_build(on: bool) -> bool:
    """Weights of +-1: pygsd_magop_unit_signed (default) or the two-stage pipeline (PYGSD_SIGNED_UNIT_BUILD=0) -- measurement /
    A-B.  Returns the previous setting."""
    global _SIGNED_UNIT_BUILD
    prev, _SIGNED_UNIT_BUILD = _SIGNED_UNIT_BUILD, bool(on)
    return prev


def set_unit_build(on: bool) -> bool:
    """Unweighted graphs: pygsd_magop_unit (default; inside the library the bucket split, or the radix-sort form with
    PYGSD_UNIT_BUILD_FORM=sort) or the two-stage pipeline (pygsd_magop_stage1 / _stage2; PYGSD_TWO_STAGE_BUILD=1) --
    measurement / A-B.  Returns the previous setting."""
    global _UNIT_BUILD
    prev, _UNIT_BUILD = _UNIT_BUILD, bool(on)
    return prev


def _unit_operator_csr(row: Tensor, col: Tensor, e: int, n: int, sym: int, q: float, lambda_max: float, diag_shift: float,
                       w: Optional[Tensor] = None, signed: bool = False, absolute_degree: bool = True):
    """pygsd_magop_unit: edge list without weights -> final CSR + values (unit weights: degrees from the row bounds, merged
    rows parked as 8-byte records between the merge and the write kernel).  ONE host read -- E_s, the bad-id witness and the
    over-long-row count -- queued between the library's two parts, so the host is back before the write kernel has finished
    and the layer's next launches queue behind it without a gap.  None: a row longer than the kernel takes (caller: two-stage
    pipeline).
    w given (round 5): pygsd_magop_unit_signed -- the same build for weights that are all +-1 (validated on the device; None
    when they are not, or when -1 meets a degree convention other than the signed Laplacian's absolute one)."""
    from ..sparse import CSR
    dev = row.device
    lib = _cabi.lib()
    with _cabi.on_device(dev):
        need = ctypes.c_size_t(0)
        check(lib.pygsd_magop_workspace(e, n, 0, ctypes.byref(need)), "pygsd_magop_workspace")
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        info = torch.empty(4, dtype=torch.int64, device=dev)
        rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        deg = torch.empty(n, dtype=torch.float32, device=dev)
        cap = 2 * e + n                                   # upper bound: every listed edge distinct, no self loops
        ccol = torch.empty(max(cap, 4), dtype=torch.int32, device=dev)
        pad = max((cap + 3) // 4 * 4, 4)
        vals = torch.empty((4, pad), dtype=torch.float32, device=dev)
        tail = (sym, float(q), float(lambda_max), float(diag_shift), ptr(ws), need.value, ptr(rowptr),
                ptr(deg), ptr(ccol), ptr(vals[0]), ptr(vals[1]), ptr(vals[2]), ptr(vals[3]), ptr(info))
        if w is None:
            fn, name, args = lib.pygsd_magop_unit, "pygsd_magop_unit", (ptr(row), ptr(col), e, n) + tail
        else:
            fn, name = lib.pygsd_magop_unit_signed, "pygsd_magop_unit_signed"
            args = (ptr(row), ptr(col), ptr(w), e, n, 1 if signed else 0, 1 if absolute_degree else 0) + tail
        check(fn(*args, 1, stream_ptr()), name)          # everything up to the row pointer
        host_info, ready = _queue_info_read(info, dev)   # behind the first part; the host waits for THIS, not for the write
        check(fn(*args, 2, stream_ptr()), name)          # the kernel that writes the slots
        ready.synchronize()                           # the one host round-trip: over while 40 % of the build is still running
        es, too_long, bad, bad_id = host_info.tolist()
    if bad:
        raise IndexError(f"edge_index holds node id {bad_id}, outside [0, {n}); the HIP path gathers and "
                         "scatters rows by these ids")
    if too_long:
        return None
    nnz = es + n
    if nnz < 0.9 * cap:
        ccol = ccol[:nnz].clone()
        tight = torch.empty((4, max((nnz + 3) // 4 * 4, 4)), dtype=torch.float32, device=dev)
        tight[:, :nnz] = vals[:, :nnz]
        vals = tight
    else:
        ccol = ccol[:nnz]
    csr = CSR(n, n, nnz, rowptr, ccol, None)
    # every row this build takes has <= 512 stream entries (pygsd_magop_unit's bound), i.e. <= 513 slots < PYGSD_LONG_ROW: no hub
    # rows, known without the device->host read CSR.hubs() would make on its first use (one per uncached forward)
    csr._hubs = ()
    return csr, (vals[2, :nnz], vals[3, :nnz]), (vals[0, :nnz], vals[1, :nnz]), deg


def fused_operator_csr(edge_index: Tensor, edge_weight: Optional[Tensor], n: int, signed: bool,
                       absolute_degree: bool, q: float, normalization: Optional[str], lambda_max: float,
                       diag_shift: float = -1.0, site=None):
    """Edge list -> compute layout of 2 L / lambda_max + diag_shift I in one pass through csrc/magop.hip
    (pygsd_magop_stage1 / _stage2): what `laplacian_parts` -> `laplacian_values(mirror=True)` ->
    `assemble_operator_csr` produce, without their int64 COO intermediates and with the node-id range check and
    the size read folded into ONE device -> host read.  For a fixed q and weights without gradient (the case the
    layers rebuild on every uncached forward, MagNetConv.py:157-181).

    Returns (CSR, (vf_real, vf_imag), (vb_real, vb_imag), deg), or None when a node has more than 4096 symmetrised
    entries (the caller then takes the generic pipeline, which has a path for such rows).
    site: the calling layer; it remembers (as `_pm1_refusals`) how often in a row its weights were not +-1."""
    from ..sparse import CSR
    _cabi.require_gpu(edge_index, edge_weight)
    if edge_index.dtype != torch.int64:
        raise TypeError("edge_index must be int64 (torch.long)")
    dev = edge_index.device
    row, col = edge_index[0].contiguous(), edge_index[1].contiguous()
    e = row.numel()
    w = None
    if edge_weight is not None:
        w = edge_weight.detach().reshape(-1).contiguous()
        if w.dtype != torch.float32:
            w = w.float()
        if w.numel() != e:
            raise ValueError(f"edge_weight has {w.numel()} entries for {e} edges")
        if e == 0:
            w = None                    # an empty tensor has no address: both stages must agree on the layout
    sym = 1 if normalization is not None else 0
    lib = _cabi.lib()
    if w is None and _UNIT_BUILD:
        built = _unit_operator_csr(row, col, e, n, sym, q, lambda_max, diag_shift)
        if built is not None:
            return built
        # a node with more than 512 symmetrised entries: the two-stage pipeline below has a path for rows up to 4096
    elif (w is not None and _UNIT_BUILD and _SIGNED_UNIT_BUILD and getattr(site, "_pm1_refusals", 0) < _PM1_SITE_REFUSALS
          and _NOT_PM1.get((edge_weight,), (signed, absolute_degree)) is None):
        # weights that are all +-1 (SDSBM / SSBM signs, explicit unit weights) take the one-call build as well; the DEVICE decides
        # (its first kernel validates the weights -- no host read of them), and a weight tensor that was turned down once is not
        # offered again while it is the same tensor at the same version (memo.TensorMemo)
        built = _unit_operator_csr(row, col, e, n, sym, q, lambda_max, diag_shift, w, signed, absolute_degree)
        if built is not None:
            if site is not None:
                site._pm1_refusals = 0
            return built
        _NOT_PM1.put((edge_weight,), (signed, absolute_degree), True)
        if site is not None:                     # `site`: the layer this build is for (any object that takes an attribute)
            site._pm1_refusals = getattr(site, "_pm1_refusals", 0) + 1
    with _cabi.on_device(dev):
        need = ctypes.c_size_t(0)
        check(lib.pygsd_magop_workspace(e, n, 0 if w is None else 1, ctypes.byref(need)), "pygsd_magop_workspace")
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        info = torch.empty(4, dtype=torch.int64, device=dev)
        rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        deg = torch.empty(n, dtype=torch.float32, device=dev)
        # E_s is not known on the host yet -- and is not needed to LAUNCH the second stage: the outputs are allocated
        # at their upper bound (every listed edge distinct, no self loops: 2 E + n; the bound is tight on the graphs this
        # is built for) and narrowed after the one host read below, which then overlaps the second stage instead of
        # draining the device between the two.  A bad id / an over-long row leaves the second stage harmless (such
        # entries were dropped / marked in the first).
        cap = 2 * e + n
        ccol = torch.empty(cap, dtype=torch.int32, device=dev)
        pad = max((cap + 3) // 4 * 4, 4)              # every value array starts on a 16-byte boundary (dwordx4 stores)
        vals = torch.empty((4, pad), dtype=torch.float32, device=dev)
        # Round 5: a weighted graph first goes through the stage whose front is the bucket split (no global sort).  It takes what
        # cannot show the order its LDS atomics deliver the entries in -- runs of at most two entries per neighbour, rows of at most
        # 512 -- and reports anything else in info[1]; the stage behind the radix sort then builds the graph (and the pair of
        # tensors is remembered, so the next build goes there directly).
        memo_key = (signed, absolute_degree)
        sorted_first = w is None or _NOT_BUCKETS.get((edge_index, edge_weight), memo_key) is not None
        for attempt in ((True,) if sorted_first else (False, True)):
            stage1 = lib.pygsd_magop_stage1_sorted if attempt else lib.pygsd_magop_stage1
            check(stage1(ptr(row), ptr(col), ptr(w), e, n, 1 if signed else 0, 1 if absolute_degree else 0,
                         sym, ptr(ws), need.value, ptr(rowptr), ptr(deg), ptr(info), stream_ptr()),
                  "pygsd_magop_stage1_sorted" if attempt else "pygsd_magop_stage1")
            host_info, ready = _queue_info_read(info, dev)   # queued behind stage 1; the host waits for THIS, not for stage 2
            check(lib.pygsd_magop_stage2(e, n, 0 if w is None else 1, float(q), sym, float(lambda_max), float(diag_shift),
                                         ptr(ws), need.value, ptr(rowptr), ptr(deg), ptr(ccol), ptr(vals[0]), ptr(vals[1]),
                                         ptr(vals[2]), ptr(vals[3]), stream_ptr()), "pygsd_magop_stage2")
            ready.synchronize()                                       # the one host round-trip
            es, too_long, bad, bad_id = host_info.tolist()
            if bad:
                raise IndexError(f"edge_index holds node id {bad_id}, outside [0, {n}); the HIP path gathers and "
                                 "scatters rows by these ids")
            if not too_long:
                break
            if not attempt:
                _NOT_BUCKETS.put((edge_index, edge_weight), memo_key, True)
        if too_long:
            return None
        nnz = es + n
        if nnz < 0.9 * cap:
            # many duplicates / self loops: the operator is memoised / cached by the layers, so views into the
            # upper-bound allocations would pin up to twice the memory it needs for its lifetime -- right-size them
            ccol = ccol[:nnz].clone()
            tight = torch.empty((4, max((nnz + 3) // 4 * 4, 4)), dtype=torch.float32, device=dev)   # rows stay 16-byte aligned
            tight[:, :nnz] = vals[:, :nnz]
            vals = tight
        else:
            ccol = ccol[:nnz]
    csr = CSR(n, n, nnz, rowptr, ccol, None)
    return csr, (vals[2, :nnz], vals[3, :nnz]), (vals[0, :nnz], vals[1, :nnz]), deg
