"""Host side of the hot path: CSR operators living in HBM and the autograd wrappers around the
C-ABI SpMM / SDDMM entry points (include/pygsd_hip.h).

Data layout in HBM (per operator):
  * `Pattern`  -- the sparsity structure of one COO `edge_index`, grouped both ways:
        fwd: CSR by OUTPUT row  (rowptr int32[n_out+1], col int32[nnz] = input row ids,
                                 perm int32[nnz] = COO entry id of each CSR slot)
        bwd: CSR by INPUT row   (same arrays with the roles swapped) -- used for dX = S^T dY
    Grouping is stable, so the entries of one output row keep their COO order: the summation order
    of the reference's scatter_add_.
  * edge values stay in COO order in the caller's tensors; `Pattern.values_for` re-orders them once
    per (tensor, version) into CSR order (fp32) and caches the copy.
  * feature matrices are fp32 row-major [N, F] with an arbitrary row stride (column slices are
    passed without a copy).

Everything here runs on the GPU through libpygsd_hip.so; CPU tensors are rejected.
"""
import ctypes
import weakref
from ctypes import c_void_p
from typing import Optional, Tuple

import torch

from . import _cabi, memo
from ._cabi import check, ptr, stream_ptr
from .memo import TensorMemo

_COLBLOCK_BYTES = 768 << 20  # dual-operator gathered set above which _spmm2_raw column-blocks

Tensor = torch.Tensor


def _rows(t: Tensor) -> Tuple[Tensor, int]:
    """Return (tensor with unit inner stride, row stride in elements)."""
    if t.dim() != 2:
        raise NotImplementedError(f"the HIP kernels take 2-D [N, F] feature matrices; got shape {tuple(t.shape)} "
                                  "(sparse.spmm folds leading batch dimensions into F; this raw entry does not)")
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"the HIP path stores features as float32 or bfloat16 (fp32 accumulate); got {t.dtype}")
    if t.size(1) > 0 and (t.stride(1) != 1 or (t.size(0) > 1 and t.stride(0) < t.size(1))):
        t = t.contiguous()
    ld = t.stride(0) if t.size(0) > 1 else max(t.size(1), 1)
    return t, max(ld, t.size(1), 1)


class CSR:
    """One orientation of a pattern: int32 rowptr / col / perm on the device."""
    __slots__ = ("n_rows", "n_cols", "nnz", "rowptr", "col", "perm", "_hubs")

    def __init__(self, n_rows, n_cols, nnz, rowptr, col, perm):
        self.n_rows, self.n_cols, self.nnz = n_rows, n_cols, nnz
        self.rowptr, self.col, self.perm = rowptr, col, perm
        self._hubs = None

    def hubs(self):
        """(row ids int32 [n_long], longest row's entry count) of the rows with more than
        PYGSD_LONG_ROW entries, or None.  Found once per CSR (one device->host read)."""
        if self._hubs is None:
            if self.nnz <= _cabi.LONG_ROW or self.n_rows == 0:
                self._hubs = ()
            else:
                deg = self.rowptr[1:] - self.rowptr[:-1]
                top = int(deg.max())
                self._hubs = () if top <= _cabi.LONG_ROW else (
                    torch.nonzero(deg > _cabi.LONG_ROW).flatten().to(torch.int32), top)
        return self._hubs or None


def _long_rows_arg(csr: CSR, n_feat: int, dual: bool):
    """(pointer-or-None, keepalive) for the long_rows argument of the fp32 SpMM entry points."""
    hubs = csr.hubs()
    if hubs is None:
        return None, None
    rows, top = hubs
    need = ctypes.c_int64(0)
    check(_cabi.lib().pygsd_spmm_long_rows_workspace(rows.numel(), top, n_feat, 1 if dual else 0,
                                                     ctypes.byref(need)), "pygsd_spmm_long_rows_workspace")
    ws = torch.empty(max(need.value, 16), dtype=torch.uint8, device=rows.device)
    desc = _cabi.LongRows(ptr(rows), rows.numel(), top, ptr(ws), need.value)
    return ctypes.byref(desc), (desc, ws)


def segment_long_rows_arg(csr: "CSR"):
    """(pointer-or-None, keepalive) for the long_rows argument of the attention / segment entry points."""
    hubs = csr.hubs()
    if hubs is None:
        return None, None
    rows, top = hubs
    need = ctypes.c_int64(0)
    check(_cabi.lib().pygsd_segment_long_rows_workspace(rows.numel(), top, ctypes.byref(need)),
          "pygsd_segment_long_rows_workspace")
    ws = torch.empty(max(need.value, 16), dtype=torch.uint8, device=rows.device)
    desc = _cabi.LongRows(ptr(rows), rows.numel(), top, ptr(ws), need.value)
    return ctypes.byref(desc), (desc, ws)


def csr_from_coo(seg: Tensor, other: Tensor, n_seg: int, n_other: int, validate: bool = True) -> CSR:
    """Group COO entries by `seg` (stable) on the device -> CSR.  seg/other: int64 [nnz].
    validate: ids outside [0, n_seg) / [0, n_other) raise IndexError (one device->host read per build) instead
    of producing a CSR whose SpMM gathers out of bounds; pass False only for ids already checked."""
    _cabi.require_gpu(seg, other)
    if seg.dtype != torch.int64 or other.dtype != torch.int64:
        raise TypeError("edge_index must be int64 (torch.long)")
    seg, other = seg.contiguous(), other.contiguous()
    if validate:
        _cabi.check_node_ids((n_seg, seg), (n_other, other))
    nnz = seg.numel()
    dev = seg.device
    lib = _cabi.lib()
    with _cabi.on_device(dev):
        rowptr = torch.empty(n_seg + 1, dtype=torch.int32, device=dev)
        col = torch.empty(nnz, dtype=torch.int32, device=dev)
        perm = torch.empty(nnz, dtype=torch.int32, device=dev)
        need = ctypes.c_size_t(0)
        check(lib.pygsd_csr_from_coo_workspace(nnz, n_seg, ctypes.byref(need)), "pygsd_csr_from_coo_workspace")
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        check(lib.pygsd_csr_from_coo(ptr(seg), ptr(other), nnz, n_seg, ptr(rowptr), ptr(col), ptr(perm),
                                     ptr(ws), need.value, stream_ptr()), "pygsd_csr_from_coo")
    return CSR(n_seg, n_other, nnz, rowptr, col, perm)


def segment_sum_raw(rowptr: Tensor, perm: Optional[Tensor], w: Tensor, n_rows: int, csr: Optional["CSR"] = None) -> Tensor:
    """out[r] = sum of w[perm[slot]] over CSR row r (pygsd_segment_sum_f32; perm None: w is in slot order).
    csr: the CSR `rowptr` belongs to -- its (cached) hub-row list routes rows with > PYGSD_LONG_ROW entries to the
    segment-parallel path."""
    out = torch.zeros(n_rows, dtype=torch.float32, device=w.device)
    if w.numel() and n_rows:
        with _cabi.on_device(w.device):
            hubs, keep = segment_long_rows_arg(csr) if csr is not None else (None, None)
            check(_cabi.lib().pygsd_segment_sum_f32(ptr(rowptr), ptr(perm), ptr(w), n_rows, ptr(out), hubs, stream_ptr()),
                  "pygsd_segment_sum_f32")
            del keep
    return out


def gather_values(src: Tensor, perm: Tensor) -> Tensor:
    """out[i] = src[perm[i]] on the device (fp32)."""
    _cabi.require_gpu(src, perm)
    src = src.contiguous()
    if src.dtype != torch.float32:
        src = src.float()
    out = torch.empty(perm.numel(), dtype=torch.float32, device=src.device)
    with _cabi.on_device(src.device):
        check(_cabi.lib().pygsd_gather_f32(ptr(src), ptr(perm), perm.numel(), ptr(out), stream_ptr()),
              "pygsd_gather_f32")
    return out


def coo_from_csr(rowptr: Tensor, col: Tensor, perm: Tensor, dtype=torch.int64) -> Tuple[Tensor, Tensor]:
    """Invert `csr_from_coo`: the (other, seg) id lists in their ORIGINAL COO order from a CSR grouped by `seg`
    (slot k holds COO entry perm[k]: other id col[k], seg id = the row whose [rowptr[r], rowptr[r+1]) holds k)."""
    nnz = col.numel()
    n_rows = rowptr.numel() - 1
    counts = (rowptr[1:] - rowptr[:-1]).long()
    row_of_slot = torch.repeat_interleave(torch.arange(n_rows, dtype=dtype, device=col.device), counts,
                                          output_size=nnz)
    other = torch.empty(nnz, dtype=dtype, device=col.device)
    seg = torch.empty(nnz, dtype=dtype, device=col.device)
    p = perm.long()
    other[p] = col.to(dtype)
    seg[p] = row_of_slot
    return other, seg


class Pattern:
    """Sparsity structure of out[scatter[e]] (+)= w[e] * x[gather[e]] for one COO edge_index.

    flow = 'source_to_target': gather = edge_index[0], scatter = edge_index[1]  (PyG default)
    flow = 'target_to_source': gather = edge_index[1], scatter = edge_index[0]  (Conv_Base)
    """

    def __init__(self, edge_index: Tensor, n_in: int, n_out: int, flow: str = "source_to_target",
                 validate: bool = True):
        """validate=False: the caller derived `edge_index` from ids that were range-checked already (the output of
        gcn_norm / conv_norm_rw / the operator builds) -- skips the device -> host read of the node-id check, the only
        synchronisation of a pattern build."""
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError(f"edge_index must be [2, E], got {tuple(edge_index.shape)}")
        if flow not in ("source_to_target", "target_to_source"):
            raise ValueError(f"unknown flow {flow!r}")
        _cabi.require_gpu(edge_index)
        g, s = (0, 1) if flow == "source_to_target" else (1, 0)
        self.n_in, self.n_out, self.nnz = int(n_in), int(n_out), int(edge_index.size(1))
        self.device = edge_index.device
        # The pattern does NOT keep the caller's edge_index (nor views of it) alive: everything later derived
        # from the COO list -- the by-source CSR, the int32 COO operands of the SDDMM, per-entry degrees -- is
        # rebuilt from the by-target CSR (`coo_from_csr`), so a cached Pattern never pins the user's tensor.
        self.fwd = csr_from_coo(edge_index[s], edge_index[g], self.n_out, self.n_in, validate=validate)
        self._bwd: Optional[CSR] = None
        self._coo32: Optional[Tuple[Tensor, Tensor]] = None
        self._inv_deg: Optional[Tensor] = None
        self._vcache = {}

    def __getstate__(self):
        # the per-tensor value cache holds weak references to caller tensors: a copied / pickled pattern starts without
        st = self.__dict__.copy()
        st["_vcache"] = {}
        return st

    @property
    def bwd(self) -> CSR:
        if self._bwd is None:
            gather, scatter = coo_from_csr(self.fwd.rowptr, self.fwd.col, self.fwd.perm, torch.int64)
            self._bwd = csr_from_coo(gather, scatter, self.n_in, self.n_out, validate=False)
        return self._bwd

    @property
    def bwd_to_fwd(self) -> Tensor:
        """int32 [nnz]: for every slot of the by-source CSR, the slot of the same entry in the by-target CSR
        (per-entry quantities produced in by-target order are read through it by the by-source reductions)."""
        m = getattr(self, "_bwd_to_fwd", None)
        if m is None:
            slot_of_entry = torch.empty(self.nnz, dtype=torch.int32, device=self.device)
            slot_of_entry[self.fwd.perm.long()] = torch.arange(self.nnz, dtype=torch.int32, device=self.device)
            m = self._bwd_to_fwd = slot_of_entry[self.bwd.perm.long()].contiguous()
        return m

    @property
    def coo32(self) -> Tuple[Tensor, Tensor]:
        """(gather, scatter) row ids as int32, COO order (SDDMM operands)."""
        if self._coo32 is None:
            self._coo32 = coo_from_csr(self.fwd.rowptr, self.fwd.col, self.fwd.perm, torch.int32)
        return self._coo32

    def mean_values(self) -> Tensor:
        """Per-entry 1 / max(in-degree of the output row, 1), COO order (backward of aggr='mean')."""
        if self._inv_deg is None:
            rp = self.fwd.rowptr
            deg = (rp[1:] - rp[:-1]).clamp(min=1).to(torch.float32)
            self._inv_deg = (1.0 / deg)[self.coo32[1].long()]
        return self._inv_deg

    def values_for(self, w: Optional[Tensor], which: str) -> Optional[Tensor]:
        """`w` (COO order) re-ordered for the `which` in {'fwd','bwd'} CSR; cached per tensor version."""
        if w is None:
            return None
        if memo.verify():
            memo.check_unchanged(w)                  # strict mode: the permuted copy below is only as good as w's contents
        key = (id(w), which)
        hit = self._vcache.get(key)
        if hit is not None and hit[0]() is w and hit[1] == (w._version, memo.content_epoch()):
            memo._order_after(hit[2], (w,))          # (a copy queued on another stream: this one waits for it first)
            return hit[2].value
        if w.numel() != self.nnz:
            raise ValueError(f"edge value array has {w.numel()} entries, pattern has {self.nnz}")
        csr = self.fwd if which == "fwd" else self.bwd
        out = gather_values(w.detach().reshape(-1), csr.perm)
        if len(self._vcache) > 16:
            self._vcache.clear()
        raw = memo._raw_stream((w,))
        mark = memo._Entry(None, None, None, out, raw, None if raw is None else torch.cuda.current_stream(w.device))
        self._vcache[key] = (weakref.ref(w), (w._version, memo.content_epoch()), mark)
        return out


# ------------------------------------------------------------------------------------------------
# raw launches
# ------------------------------------------------------------------------------------------------
def _spmm_bf16_raw(csr: CSR, val: Optional[Tensor], x: Tensor, ldx: int, z: Optional[Tensor], alpha: float,
                   beta: float, mean: bool, bias: Optional[Tensor] = None) -> Tensor:
    """bf16-storage SpMM (fp32 values / accumulation).  Shapes the 16-byte-row kernel cannot address
    are widened to fp32, run through the fp32 kernel and rounded back -- still the HIP path."""
    f = x.size(1)
    ok = f % 8 == 0 and ldx % 8 == 0 and x.data_ptr() % 16 == 0
    zp, ldz = None, 0
    if z is not None:
        z, ldz = _rows(z.to(torch.bfloat16))
        ok = ok and ldz % 8 == 0 and z.data_ptr() % 16 == 0
        zp = ptr(z)
    elif bias is not None:                      # one row for every output row: Z with a zero row stride
        bias = bias.detach().to(torch.bfloat16).contiguous()
        ok = ok and bias.data_ptr() % 16 == 0
        zp, beta = ptr(bias), 1.0
    if not ok:
        y32 = _spmm_raw(csr, val, x.float(), None if z is None else z.float(), alpha, beta, mean,
                        None if bias is None else bias.float())
        return y32.to(torch.bfloat16)
    y = torch.empty((csr.n_rows, f), dtype=torch.bfloat16, device=x.device)
    with _cabi.on_device(x.device):
        check(_cabi.lib().pygsd_spmm_csr_bf16(ptr(csr.rowptr), ptr(csr.col), ptr(val), ptr(x), ldx, ptr(y), f, zp,
                                              ldz, csr.n_rows, f, float(alpha), float(beta), 1 if mean else 0,
                                              stream_ptr()), "pygsd_spmm_csr_bf16")
    return y


def _spmm_raw(csr: CSR, val: Optional[Tensor], x: Tensor, z: Optional[Tensor], alpha: float, beta: float,
              mean: bool, bias: Optional[Tensor] = None) -> Tensor:
    """bias: a [F] vector added to every output row in the kernel's epilogue (Z read with a zero row stride);
    exclusive with z."""
    _cabi.require_gpu(x, z, val, bias)
    if bias is not None and z is not None:
        raise ValueError("spmm: bias and z are exclusive")
    x, ldx = _rows(x)
    if x.dtype == torch.bfloat16 and csr.nnz > 0 and x.size(1) > 0 and csr.n_rows > 0:
        if x.size(0) != csr.n_cols:
            raise ValueError(f"x has {x.size(0)} rows, operator expects {csr.n_cols}")
        return _spmm_bf16_raw(csr, val, x, ldx, z, alpha, beta, mean, bias)
    if x.size(0) != csr.n_cols:
        raise ValueError(f"x has {x.size(0)} rows, operator expects {csr.n_cols}")
    f = x.size(1)
    if z is not None and tuple(z.shape) != (csr.n_rows, f):
        raise ValueError(f"z has shape {tuple(z.shape)}, expected {(csr.n_rows, f)}")
    if f % 4 and x.dtype == torch.float32 and csr.nnz > 0 and csr.n_rows > 0:
        # widths like a 5-class output layer: zero-pad to 16-byte rows so the vector kernel runs (the scalar
        # fallback walks a row's neighbours one by one: 2.9 ms against ~1.1 ms at 2M nodes / 52M entries / F = 5)
        pad = (0, 4 - f % 4)
        y = _spmm_raw(csr, val, torch.nn.functional.pad(x, pad), None if z is None else torch.nn.functional.pad(z, pad),
                      alpha, beta, mean, None if bias is None else torch.nn.functional.pad(bias, pad))
        return y[:, :f]
    if csr.nnz == 0 or f == 0 or csr.n_rows == 0:  # edgeless operator: nothing to gather
        y = torch.zeros((csr.n_rows, f), dtype=x.dtype, device=x.device)
        if bias is not None:
            return y.add_(bias.to(y.dtype))
        return y if z is None else y.add_(z, alpha=beta)
    y = torch.empty((csr.n_rows, f), dtype=torch.float32, device=x.device)
    zp, ldz = None, 0
    if z is not None:
        z, ldz = _rows(z)
        zp = ptr(z)
    elif bias is not None:
        bias = bias.detach().float().contiguous()
        if bias.data_ptr() % 16:
            bias = bias.clone()
        zp, beta = ptr(bias), 1.0
    with _cabi.on_device(x.device):
        hubs, keep = _long_rows_arg(csr, f, False)
        check(_cabi.lib().pygsd_spmm_csr_f32(ptr(csr.rowptr), ptr(csr.col), ptr(val), ptr(x), ldx, ptr(y),
                                             max(f, 1), zp, ldz, csr.n_rows, f, float(alpha), float(beta),
                                             1 if mean else 0, csr.nnz, hubs, stream_ptr()),
              "pygsd_spmm_csr_f32")
        del keep
    return y


def _spmm2_raw(csr: CSR, val_a: Tensor, val_b: Tensor, xa: Tensor, xb: Tensor, za: Optional[Tensor],
               zb: Optional[Tensor], alpha: float, beta: float) -> Tuple[Tensor, Tensor]:
    _cabi.require_gpu(xa, xb, za, zb, val_a, val_b)
    for t in (xa, xb, za, zb):
        if t is not None and t.dtype != torch.float32:
            raise TypeError(f"the dual-operator SpMM computes in float32; got {t.dtype}")
    if xa.shape != xb.shape:
        raise ValueError("spmm2: the two inputs must have the same shape")
    xa, lda = _rows(xa)
    xb, ldb = _rows(xb)
    if lda != ldb:
        xa, xb = xa.contiguous(), xb.contiguous()
        lda = max(xa.size(1), 1)
    if xa.size(0) != csr.n_cols:
        raise ValueError(f"x has {xa.size(0)} rows, operator expects {csr.n_cols}")
    f = xa.size(1)
    if f % 4 and csr.nnz > 0 and csr.n_rows > 0:          # see _spmm_raw: pad to 16-byte rows for the vector kernel
        pad = (0, 4 - f % 4)
        P = torch.nn.functional.pad
        ya, yb = _spmm2_raw(csr, val_a, val_b, P(xa, pad), P(xb, pad), None if za is None else P(za, pad),
                            None if zb is None else P(zb, pad), alpha, beta)
        return ya[:, :f], yb[:, :f]
    if csr.nnz == 0 or f == 0 or csr.n_rows == 0:
        ya = torch.zeros((csr.n_rows, f), dtype=torch.float32, device=xa.device)
        yb = torch.zeros_like(ya)
        return (ya, yb) if za is None else (ya.add_(za, alpha=beta), yb.add_(zb, alpha=beta))
    ya = torch.empty((csr.n_rows, f), dtype=torch.float32, device=xa.device)
    yb = torch.empty_like(ya)
    zap = zbp = None
    ldz = 0
    if za is not None:
        za, ldz = _rows(za)
        zb, ldz_b = _rows(zb)
        if ldz != ldz_b:
            za, zb = za.contiguous(), zb.contiguous()
            ldz = max(f, 1)
        zap, zbp = ptr(za), ptr(zb)
    # Column blocking: once the two gathered matrices no longer fit anywhere near the 256 MB Infinity Cache
    # (1M x 128 x 2 operands = 1 GB), two passes over 64-column halves re-read the CSR streams (+3 % bytes) but
    # halve the gathered set -- measured 6.24 -> 5.75 ms on 1M nodes / 41M entries / F=128
    # (tools/colblock_probe.py; narrower blocks or the single-operator kernel do not gain).
    blocks = [(0, f)]
    if f >= 128 and f % 64 == 0 and csr.n_cols * f * 8 >= _COLBLOCK_BYTES:
        blocks = [(o, 64) for o in range(0, f, 64)]
    with _cabi.on_device(xa.device):
        lib = _cabi.lib()
        for off, width in blocks:
            hubs, keep = _long_rows_arg(csr, width, True)
            b = 4 * off
            check(lib.pygsd_spmm2_csr_f32(ptr(csr.rowptr), ptr(csr.col), ptr(val_a), ptr(val_b),
                                          c_void_p(xa.data_ptr() + b), c_void_p(xb.data_ptr() + b), lda,
                                          c_void_p(ya.data_ptr() + b), c_void_p(yb.data_ptr() + b), max(f, 1),
                                          None if zap is None else c_void_p(za.data_ptr() + b),
                                          None if zbp is None else c_void_p(zb.data_ptr() + b), ldz,
                                          csr.n_rows, width, float(alpha), float(beta), csr.nnz, hubs,
                                          stream_ptr()),
                  "pygsd_spmm2_csr_f32")
            del keep
    return ya, yb


def _hint(csr: CSR, n_rows: int) -> int:
    """nnz_hint of a row-range launch: the kernels pick their pipelining variant from entries per LAUNCHED row."""
    return max(int(csr.nnz * (n_rows / max(csr.n_rows, 1))), 1)


def spmm2_rows_into(csr: CSR, val_a: Tensor, val_b: Tensor, xa: Tensor, xb: Tensor, ya: Tensor, yb: Tensor,
                    row_lo: int = 0, row_hi: Optional[int] = None, alpha: float = 1.0,
                    accumulate: bool = False) -> None:
    """Rows [row_lo, row_hi) of the dual product, written IN PLACE into preallocated outputs:
        ya[r] = alpha * sum_e val_a[e] xa[col[e]] (+ ya[r] if accumulate), same for b.
    The building block of the sharded, pipelined propagate (parallel.py): one launch per (column block of the
    operator, row chunk), partial products accumulated through the kernel's own beta * Z epilogue with Z = Y
    (each output row is read and written by the one wavefront that owns it).  fp32, F % 4 == 0."""
    _cabi.require_gpu(xa, xb, ya, yb, val_a, val_b)
    row_hi = csr.n_rows if row_hi is None else row_hi
    n_rows, f = row_hi - row_lo, xa.size(1)
    if n_rows <= 0 or f == 0:
        return
    if f % 4 or any(t.dtype != torch.float32 for t in (xa, xb, ya, yb)):
        raise TypeError("spmm2_rows_into: float32 features with a width that is a multiple of 4")
    xa, lda = _rows(xa)
    xb, ldb = _rows(xb)
    ldy = ya.stride(0)
    if lda != ldb or yb.stride(0) != ldy or ya.stride(1) != 1 or yb.stride(1) != 1:
        raise ValueError("spmm2_rows_into: operands of one product must share their row stride")
    if csr.nnz == 0:
        if not accumulate:
            ya[row_lo:row_hi].zero_()
            yb[row_lo:row_hi].zero_()
        return
    off_y = 4 * row_lo * ldy
    pya, pyb = c_void_p(ya.data_ptr() + off_y), c_void_p(yb.data_ptr() + off_y)
    with _cabi.on_device(xa.device):
        check(_cabi.lib().pygsd_spmm2_csr_f32(c_void_p(csr.rowptr.data_ptr() + 4 * row_lo), ptr(csr.col), ptr(val_a),
                                              ptr(val_b), ptr(xa), ptr(xb), lda, pya, pyb, ldy,
                                              pya if accumulate else None, pyb if accumulate else None,
                                              ldy if accumulate else 0, n_rows, f, float(alpha),
                                              1.0 if accumulate else 0.0, _hint(csr, n_rows), None, stream_ptr()),
              "pygsd_spmm2_csr_f32")


def spmm_rows_into(csr: CSR, val: Optional[Tensor], x: Tensor, y: Tensor, row_lo: int = 0,
                   row_hi: Optional[int] = None, alpha: float = 1.0, accumulate: bool = False,
                   mean: bool = False, z: Optional[Tensor] = None) -> None:
    """Single-operator form of `spmm2_rows_into`; float32 (F % 4 == 0) or bfloat16 storage (F % 8 == 0, fp32
    values and accumulation).  `mean` divides by the row's entry count and is only meaningful for an operator
    that is NOT split into column blocks."""
    _cabi.require_gpu(x, y, val)
    row_hi = csr.n_rows if row_hi is None else row_hi
    n_rows, f = row_hi - row_lo, x.size(1)
    if n_rows <= 0 or f == 0:
        return
    bf16 = x.dtype == torch.bfloat16
    mixed = bf16 and y.dtype == torch.float32          # bf16 gather, fp32 partial products (phased products)
    if (x.dtype != y.dtype and not mixed) or x.dtype not in (torch.float32, torch.bfloat16) or f % (8 if bf16 else 4):
        raise TypeError("spmm_rows_into: float32 (F % 4 == 0) or bfloat16 (F % 8 == 0) features; the output has the "
                        "input's dtype, or float32 for bfloat16 inputs")
    if mixed and mean:
        raise ValueError("spmm_rows_into: fp32-accumulated bf16 products do not take mean")
    x, ldx = _rows(x)
    ldy = y.stride(0)
    if y.stride(1) != 1:
        raise ValueError("spmm_rows_into: output rows must have unit inner stride")
    if csr.nnz == 0:                                     # edgeless operator: y = 0 (+ z) -- nothing to gather, no launch
        if z is not None:
            if accumulate or row_lo or z.dtype != y.dtype or z.size(0) != y.size(0):
                raise ValueError("spmm_rows_into: z comes without accumulate / row offsets and has the output's dtype and height")
            y[row_lo:row_hi].copy_(z[row_lo:row_hi, :f])
        elif not accumulate:
            y[row_lo:row_hi].zero_()
        return
    esz = 2 if (bf16 and not mixed) else 4
    py = c_void_p(y.data_ptr() + esz * row_lo * ldy)
    rp = c_void_p(csr.rowptr.data_ptr() + 4 * row_lo)
    if z is not None:
        # y = alpha * S x + z with z ANOTHER matrix of the output's dtype (own row stride; full height, row_lo = 0)
        if accumulate or row_lo or z.dtype != y.dtype or z.stride(1) != 1 or z.size(0) != y.size(0):
            raise ValueError("spmm_rows_into: z comes without accumulate / row offsets and has the output's dtype and height")
        z, ldz, beta = c_void_p(z.data_ptr()), z.stride(0), 1.0
    else:
        z, ldz, beta = (py, ldy, 1.0) if accumulate else (None, 0, 0.0)
    with _cabi.on_device(x.device):
        if mixed:
            check(_cabi.lib().pygsd_spmm_csr_bf16_acc_f32(rp, ptr(csr.col), ptr(val), ptr(x), ldx, py, ldy, z, ldz, n_rows, f,
                                                          float(alpha), beta, stream_ptr()), "pygsd_spmm_csr_bf16_acc_f32")
        elif bf16:
            check(_cabi.lib().pygsd_spmm_csr_bf16(rp, ptr(csr.col), ptr(val), ptr(x), ldx, py, ldy, z, ldz, n_rows, f,
                                                  float(alpha), beta, 1 if mean else 0, stream_ptr()),
                  "pygsd_spmm_csr_bf16")
        else:
            check(_cabi.lib().pygsd_spmm_csr_f32(rp, ptr(csr.col), ptr(val), ptr(x), ldx, py, ldy, z, ldz, n_rows, f,
                                                 float(alpha), beta, 1 if mean else 0, _hint(csr, n_rows), None,
                                                 stream_ptr()),
                  "pygsd_spmm_csr_f32")


def _sddmm_raw(ia: Tensor, ib: Tensor, a: Tensor, b: Tensor) -> Tensor:
    """out[e] = <a[ia[e]], b[ib[e]]>."""
    _cabi.require_gpu(ia, ib, a, b)
    a, lda = _rows(a.float())
    b, ldb = _rows(b.float())
    out = torch.empty(ia.numel(), dtype=torch.float32, device=a.device)
    with _cabi.on_device(a.device):
        check(_cabi.lib().pygsd_sddmm_coo_f32(ptr(ia), ptr(ib), ia.numel(), ptr(a), lda, ptr(b), ldb,
                                              a.size(1), ptr(out), stream_ptr()), "pygsd_sddmm_coo_f32")
    return out


# ------------------------------------------------------------------------------------------------
# autograd
# ------------------------------------------------------------------------------------------------
class _Spmm(torch.autograd.Function):
    """y = alpha * reduce_{e -> row} w[e] * x[gather[e]] + beta * z   (reduce = add | mean)."""

    @staticmethod
    def forward(ctx, x, w, z, pat: Pattern, alpha: float, beta: float, mean: bool):
        y = _spmm_raw(pat.fwd, pat.values_for(w, "fwd"), x, z, alpha, beta, mean)
        ctx.pat, ctx.alpha, ctx.beta, ctx.mean = pat, alpha, beta, mean
        ctx.has_w = w is not None
        ctx.save_for_backward(x if (w is not None and w.requires_grad) else None, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        pat, alpha = ctx.pat, ctx.alpha
        x_saved, w = ctx.saved_tensors
        gx = gw = gz = None
        gy = gy.contiguous()
        if ctx.needs_input_grad[0]:
            if ctx.mean:
                wb = pat.mean_values() if w is None else w * pat.mean_values()
                gx = _spmm_raw(pat.bwd, pat.values_for(wb, "bwd"), gy, None, alpha, 0.0, False)
            else:
                gx = _spmm_raw(pat.bwd, pat.values_for(w, "bwd"), gy, None, alpha, 0.0, False)
        if ctx.has_w and ctx.needs_input_grad[1]:
            gi, si = pat.coo32
            gw = _sddmm_raw(gi, si, x_saved, gy)
            if ctx.mean:
                gw = gw * pat.mean_values()
            if alpha != 1.0:
                gw = gw * alpha
            gw = gw.view_as(w).to(w.dtype)
        if ctx.needs_input_grad[2]:
            gz = gy * ctx.beta
        return gx, gw, gz, None, None, None, None


class _Spmm2(torch.autograd.Function):
    """(ya, yb) = alpha * (S_a xa, S_b xb) + beta * (za, zb); S_a, S_b share one pattern."""

    @staticmethod
    def forward(ctx, xa, xb, wa, wb, za, zb, pat: Pattern, alpha: float, beta: float):
        ya, yb = _spmm2_raw(pat.fwd, pat.values_for(wa, "fwd"), pat.values_for(wb, "fwd"), xa, xb, za, zb,
                            alpha, beta)
        ctx.pat, ctx.alpha, ctx.beta = pat, alpha, beta
        need_x = wa.requires_grad or wb.requires_grad
        ctx.save_for_backward(xa if need_x else None, xb if need_x else None, wa, wb)
        return ya, yb

    @staticmethod
    def backward(ctx, ga, gb):
        pat, alpha = ctx.pat, ctx.alpha
        xa, xb, wa, wb = ctx.saved_tensors
        ga, gb = ga.contiguous(), gb.contiguous()
        gxa = gxb = gwa = gwb = gza = gzb = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gxa, gxb = _spmm2_raw(pat.bwd, pat.values_for(wa, "bwd"), pat.values_for(wb, "bwd"), ga, gb,
                                  None, None, alpha, 0.0)
        if ctx.needs_input_grad[2]:
            gi, si = pat.coo32
            gwa = (_sddmm_raw(gi, si, xa, ga) * alpha).view_as(wa)
        if ctx.needs_input_grad[3]:
            gi, si = pat.coo32
            gwb = (_sddmm_raw(gi, si, xb, gb) * alpha).view_as(wb)
        if ctx.needs_input_grad[4]:
            gza = ga * ctx.beta
        if ctx.needs_input_grad[5]:
            gzb = gb * ctx.beta
        return gxa, gxb, gwa, gwb, gza, gzb, None, None, None


def spmm(pat: Pattern, x: Tensor, w: Optional[Tensor] = None, *, z: Optional[Tensor] = None,
         alpha: float = 1.0, beta: float = 0.0, reduce: str = "add") -> Tensor:
    """Differentiable gather-scale-reduce over `pat`: the fused replacement of
    MessagePassing.propagate for message = w * x_j."""
    if reduce not in ("add", "sum", "mean"):
        raise ValueError(f"unsupported reduce {reduce!r}")
    if x.dim() > 2:
        # [..., N, F]: the reference's node_dim = -2 propagate accepts leading batch dimensions (DGCNConv.py:83-97,
        # SGCNConv.py:101-119).  The operator acts on the node axis only, so the batch folds into the feature axis:
        # one SpMM at width B * F, gradients (also w.r.t. the edge values: summed over the batch) through the views
        lead, n, f = x.shape[:-2], x.size(-2), x.size(-1)

        def fold(t):
            return t.reshape(-1, t.size(-2), f).permute(1, 0, 2).reshape(t.size(-2), -1)
        y = spmm(pat, fold(x), w, z=None if z is None else fold(z.expand(lead + (pat.n_out, f))), alpha=alpha, beta=beta,
                 reduce=reduce)
        return y.reshape(pat.n_out, -1, f).permute(1, 0, 2).reshape(lead + (pat.n_out, f))
    return _Spmm.apply(x, w, z, pat, float(alpha), float(beta), reduce == "mean")


def spmm2(pat: Pattern, xa: Tensor, xb: Tensor, wa: Tensor, wb: Tensor, *, za: Optional[Tensor] = None,
          zb: Optional[Tensor] = None, alpha: float = 1.0, beta: float = 0.0) -> Tuple[Tensor, Tensor]:
    """Two operators on one pattern in one traversal (real / imaginary magnetic Laplacian)."""
    if (za is None) != (zb is None):
        raise ValueError("spmm2: za and zb must be given together")
    return _Spmm2.apply(xa, xb, wa, wb, za, zb, pat, float(alpha), float(beta))


# ------------------------------------------------------------------------------------------------
# x[edge_index[row]] with a segment-reduce backward
# ------------------------------------------------------------------------------------------------
class _GatherRows(torch.autograd.Function):
    """out[e] = x[index[e]].  torch's own backward of an index gather sorts the E indices on EVERY call
    (2.1 ms per gather at E = 10^7, whatever the row width); here the grouping of the positions by node comes
    from the (cached) CSR of the edge list and the backward is one value-less SpMM over the [E, F] gradient:
    dx[v] = sum of g[e] over the positions e with index[e] = v -- deterministic, no atomics."""

    @staticmethod
    def forward(ctx, x, index, csr):
        ctx.csr, ctx.width = csr, x.size(1)
        return x.index_select(0, index)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        csr, f = ctx.csr, ctx.width
        g = g.float()
        if f % 4:                                   # 16-byte rows for the vector kernel
            g = torch.nn.functional.pad(g, (0, 4 - f % 4))
        by_node = CSR(csr.n_rows, csr.nnz, csr.nnz, csr.rowptr, csr.perm, None)
        by_node._hubs = csr.hubs() or ()            # same row lengths: reuse the (cached) hub-row scan
        dx = _spmm_raw(by_node, None, g.contiguous(), None, 1.0, 0.0, False)
        return dx[:, :f], None, None


def gather_rows(x: Tensor, edge_index: Tensor, row: int, cached: bool = True) -> Tensor:
    """x[edge_index[row]] for a [2, E] edge list, differentiable w.r.t. x through the edge list's CSR
    (`cached=False` for one-off index sets such as freshly sampled negatives: they skip the pattern cache)."""
    _cabi.require_gpu(x, edge_index)
    n = x.size(0)
    if edge_index.size(1) == 0 or not x.requires_grad:
        return x.index_select(0, edge_index[row])
    pat = GLOBAL_PATTERNS.get(edge_index, n, n, "source_to_target") if cached else Pattern(edge_index, n, n)
    return _GatherRows.apply(x, edge_index[row], pat.fwd if row == 1 else pat.bwd)


# ------------------------------------------------------------------------------------------------
# pattern cache for raw-tensor callers (MessagePassing.propagate with a plain edge_index)
# ------------------------------------------------------------------------------------------------
class PatternCache:
    """LRU of the last few patterns, keyed on the edge_index TENSOR OBJECT (held weakly), its in-place version
    and storage (memo.TensorMemo).  Pure function of the tensor's content: never changes results, only avoids
    re-sorting when a caller passes the same edge_index again (e.g. SIMPA's 7 calls).  Neither the cache nor
    the patterns in it keep the caller's tensor alive; an entry goes away with its edge_index.  See memo.py for
    the invalidation contract and the opt-outs (PYGSD_NO_OPERATOR_MEMO=1, memo.set_enabled(False))."""

    def __init__(self, capacity: int = 8):
        self._memo = TensorMemo(capacity)

    def get(self, edge_index: Tensor, n_in: int, n_out: int, flow: str, validate: bool = True, trusted: bool = False) -> Pattern:
        key = (int(n_in), int(n_out), flow)
        pat = self._memo.get((edge_index,), key, trusted=trusted)
        if pat is None:
            pat = self._memo.put((edge_index,), key, Pattern(edge_index, n_in, n_out, flow, validate))
        return pat

    def clear(self):
        self._memo.clear()

    def __len__(self):
        return len(self._memo)


GLOBAL_PATTERNS = PatternCache()
