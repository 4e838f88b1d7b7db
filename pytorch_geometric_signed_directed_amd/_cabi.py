"""ctypes binding of libpygsd_hip.so (include/pygsd_hip.h).  No torch types cross this boundary:
tensors are handed over as raw device addresses (`tensor.data_ptr()`), the stream as the raw
hipStream_t of torch's current stream.

The library is REQUIRED: there is no CPU or eager-PyTorch fallback behind these functions.  If the
shared object is missing or fails to load, importing an op raises immediately.
"""
import collections
import ctypes
import os
import threading
import warnings
from ctypes import c_double, c_float, c_int32, c_int64, c_size_t, c_void_p

import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libpygsd_hip.so")
_lib = None

K_SPMM, K_SPMM2, K_SDDMM, K_BUILD, K_ELEMENTWISE, K_DENSE, K_DENSE_BWD = range(7)
KERNEL_IDS = {"spmm": K_SPMM, "spmm2": K_SPMM2, "sddmm": K_SDDMM, "build": K_BUILD,
              "elementwise": K_ELEMENTWISE, "dense": K_DENSE, "dense_bwd": K_DENSE_BWD}

LONG_ROW = 4096  # PYGSD_LONG_ROW


class LongRows(ctypes.Structure):
    """struct pygsd_long_rows (include/pygsd_hip.h)."""
    _fields_ = [("rows", c_void_p), ("n_rows", c_int32), ("max_entries", c_int32), ("workspace", c_void_p),
                ("workspace_bytes", c_int64)]


# name -> (restype, argtypes); must list every symbol include/pygsd_hip.h declares
PROTOTYPES = {
    "pygsd_version": (c_int32, []),
    "pygsd_last_error": (ctypes.c_char_p, []),
    "pygsd_spmm_csr_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                     c_void_p, c_int64, c_int32, c_int32, c_float, c_float, c_int32,
                                     c_int64, c_void_p, c_void_p]),
    "pygsd_spmm_long_rows_workspace": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "pygsd_spmm_csr_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                      c_void_p, c_int64, c_int32, c_int32, c_float, c_float, c_int32,
                                      c_void_p]),
    "pygsd_spmm_csr_bf16_acc_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                              c_void_p, c_int64, c_int32, c_int32, c_float, c_float, c_void_p]),
    "pygsd_spmm2_csr_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                      c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32,
                                      c_int32, c_float, c_float, c_int64, c_void_p, c_void_p]),
    "pygsd_spmm2_k1_dense_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                           c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int64, c_void_p]),
    "pygsd_sddmm_coo_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                      c_int32, c_void_p, c_void_p]),
    # the attention / segment entry points take (..., long_rows descriptor or NULL, stream) last
    "pygsd_segment_long_rows_workspace": (c_int32, [c_int32, c_int32, c_void_p]),
    "pygsd_gat_alpha_csr_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_float, c_void_p,
                                          c_void_p, c_void_p]),
    "pygsd_segment_softmax_csr_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "pygsd_segment_softmax_bwd_csr_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                                    c_void_p]),
    "pygsd_gat_alpha_bwd_csr_v2_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                                 c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32,
                                                 c_void_p, c_void_p, c_void_p, c_void_p]),
    "pygsd_segment_sum_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "pygsd_snea_alpha_csr_f32": (c_int32, [c_void_p] * 8 + [c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                                            c_void_p]),
    "pygsd_snea_alpha_bwd_csr_f32": (c_int32, [c_void_p] * 11 + [c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                                                c_void_p, c_void_p]),
    "pygsd_gat_alpha_bwd_csr_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                              c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int32,
                                              c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pygsd_csr_from_coo_workspace": (c_int32, [c_int64, c_int32, ctypes.POINTER(c_size_t)]),
    "pygsd_csr_from_coo": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
    "pygsd_gather_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "pygsd_sort_keys_u64_workspace": (c_int32, [c_int64, ctypes.POINTER(c_size_t)]),
    "pygsd_sort_keys_u64": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_size_t,
                                      c_void_p]),
    "pygsd_maglap_workspace": (c_int32, [c_int64, ctypes.POINTER(c_size_t)]),
    "pygsd_maglap_sort": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_size_t, c_void_p, c_void_p]),
    "pygsd_maglap_merge": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int64, c_void_p, c_size_t,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pygsd_maglap_assemble_csr": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_int64, c_int32, c_float, c_float, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pygsd_maglap_values": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double,
                                      c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pygsd_magop_workspace": (c_int32, [c_int64, c_int32, c_int32, ctypes.POINTER(c_size_t)]),
    "pygsd_magop_stage1": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                     c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pygsd_magop_stage1_sorted": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                            c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pygsd_magop_stage2": (c_int32, [c_int64, c_int32, c_int32, c_double, c_int32, c_float, c_float, c_void_p, c_size_t,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pygsd_magop_unit": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_double, c_float, c_float, c_void_p, c_size_t,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "pygsd_magop_unit_signed": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_double, c_float,
                                          c_float, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int32, c_void_p]),
    "pygsd_self_loops_workspace": (c_int32, [c_int64, ctypes.POINTER(c_size_t)]),
    "pygsd_self_loops_scan": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_size_t, c_void_p,
                                        c_void_p, c_void_p]),
    "pygsd_self_loops_emit": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_int64,
                                        c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pygsd_csr_row_sum_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "pygsd_degree_scale_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p,
                                         c_void_p]),
    "pygsd_complex_relu_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "pygsd_complex_relu_bwd_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                             c_void_p]),
    "pygsd_magnetic_dense_supported": (c_int32, [c_int32, c_int32, c_int32]),
    "pygsd_magnetic_dense_fwd_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "pygsd_magnetic_dense_bwd_workspace": (c_int32, [c_int32, c_int32, c_int32, c_int32,
                                                     ctypes.POINTER(c_size_t)]),
    "pygsd_magnetic_dense_bwd_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int64,
                                               c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                               c_int32, c_void_p, c_size_t, c_void_p]),
    "pygsd_magnetic_dense_fwd_pieces_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                                      c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "pygsd_magnetic_dense_bwd_pieces_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int64,
                                                      c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                                      c_int32, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "pygsd_gather_pieces_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                          c_void_p]),
    "pygsd_id_range_i64": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p]),
    "pygsd_stream_copy_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "pygsd_spin_us": (c_int32, [c_double, c_void_p]),
    "pygsd_pack_slices": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int64, c_int32, c_int32, c_int32, c_void_p,
                                    c_void_p]),
    "pygsd_weighted_sum_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p, c_int64, c_void_p]),
    "pygsd_dots_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p, c_size_t,
                                 c_void_p]),
    "pygsd_tall_linear_supported": (c_int32, [c_int32, c_int32, c_int32]),
    "pygsd_tall_f32_form": (c_int32, [c_int32]),
    "pygsd_dense_f32_form": (c_int32, [c_int32]),
    "pygsd_tall_linear": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p]),
    "pygsd_column_sums_workspace": (c_int32, [c_int64, c_int32, c_int32, ctypes.POINTER(ctypes.c_size_t)]),
    "pygsd_column_sums": (c_int32, [c_void_p, c_int64, c_int64, c_int32, c_int32, c_void_p, c_void_p, ctypes.c_size_t,
                                    c_void_p]),
    "pygsd_tall_gram_workspace": (c_int32, [c_int64, c_int32, c_int32, c_int32, ctypes.POINTER(ctypes.c_size_t)]),
    "pygsd_tall_gram": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int64,
                                  c_int32, c_void_p, c_void_p, ctypes.c_size_t, c_void_p]),
    "pygsd_gemm_f32_workspace": (c_int32, [c_int64, c_int64, c_int64, ctypes.POINTER(ctypes.c_size_t)]),
    "pygsd_gemm_f32": (c_int32, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64,
                                 c_int64, c_int64, c_int32, c_void_p, ctypes.c_size_t, c_void_p]),
    "pygsd_gemm_bf16": (c_int32, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int32,
                                  c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, ctypes.c_size_t, c_void_p]),
    "pygsd_fingerprint_u64": (c_int32, [c_void_p, ctypes.c_size_t, c_void_p, c_void_p]),
    "pygsd_prof_enable": (c_int32, [c_int32]),
    "pygsd_prof_reset": (c_int32, []),
    "pygsd_prof_collect": (c_int32, [c_int32, ctypes.POINTER(c_int64), ctypes.POINTER(c_double)]),
}
ABI_VERSION = 17


class PieceLayoutStruct(ctypes.Structure):
    """include/pygsd_hip.h: pygsd_piece_layout."""
    _fields_ = [("base", c_int64 * 4), ("lo", c_int32 * 5), ("rows", c_int32 * 4), ("n_chunks", c_int32),
                ("blk_rows", c_int32), ("slots_per_blk", c_int32), ("row_stride", c_int32), ("slot_floats", c_int32),
                ("replicas", c_int32)]


class PieceLayout:
    """Host description of a piece layout (include/pygsd_hip.h, pygsd_piece_layout): where element (row t, column c) of an
    [n_rows, F] operand lives in the send buffers of an inbound exchange / the receive buffer of a return exchange.

        blk = t // blk_rows, u = t % blk_rows, r: lo[r] <= u < lo[r + 1], j = c // slot_floats
        slot = (blk + replica) * slots_per_blk + j
        offset = base[r] + (slot * rows[r] + (u - lo[r])) * row_stride + c % slot_floats         (elements from the operand pointer)

    `offsets` is the restatement of that formula in tensor ops: the CPU tests hold the layouts the engine builds to the
    tensor-op packing / merging with it, and the GPU tests hold the kernels to it."""

    def __init__(self, base, lo, rows, blk_rows, slots_per_blk, row_stride, slot_floats, replicas=1):
        self.base, self.lo, self.rows = [int(b) for b in base], [int(v) for v in lo], [int(v) for v in rows]
        self.blk_rows, self.slots_per_blk = int(blk_rows), int(slots_per_blk)
        self.row_stride, self.slot_floats, self.replicas = int(row_stride), int(slot_floats), int(replicas)
        if not (1 <= len(self.rows) <= 4 and len(self.lo) == len(self.rows) + 1 and len(self.base) == len(self.rows)):
            raise ValueError("a piece layout holds 1..4 chunks")

    def struct(self) -> PieceLayoutStruct:
        n = len(self.rows)
        s = PieceLayoutStruct()
        for r in range(4):
            s.base[r] = self.base[r] if r < n else 0
            s.rows[r] = self.rows[r] if r < n else 0
        for r in range(5):
            s.lo[r] = self.lo[r] if r <= n else self.lo[n]
        s.n_chunks, s.blk_rows, s.slots_per_blk = n, self.blk_rows, self.slots_per_blk
        s.row_stride, s.slot_floats, s.replicas = self.row_stride, self.slot_floats, self.replicas
        return s

    def offsets(self, n_rows: int, width: int, replica: int = 0):
        """int64 [n_rows, width]: element offset of every (row, column)."""
        t = torch.arange(n_rows, dtype=torch.long).view(-1, 1)
        c = torch.arange(width, dtype=torch.long).view(1, -1)
        blk, u = t // self.blk_rows, t % self.blk_rows
        inner = torch.tensor(self.lo[1:-1], dtype=torch.long)
        r = torch.searchsorted(inner, u.view(-1), right=True).view(-1, 1) if inner.numel() else torch.zeros_like(u)
        lo = torch.tensor(self.lo[:-1], dtype=torch.long)[r]
        rows = torch.tensor(self.rows, dtype=torch.long)[r]
        base = torch.tensor(self.base, dtype=torch.long)[r]
        slot = (blk + replica) * self.slots_per_blk + c // self.slot_floats
        return base + (slot * rows + (u - lo)) * self.row_stride + c % self.slot_floats


def lib_path():
    return _LIB_PATH


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        # a source-only checkout: compile in-tree once (hipcc, gfx950); never fall back to anything else
        try:
            from .build import build_library
            build_library()
        except Exception as exc:  # noqa: BLE001
            raise RuntimeError(
                f"pytorch_geometric_signed_directed_amd: HIP library not built ({_LIB_PATH} is missing) and "
                f"building it failed ({exc}). Run `python -c 'import __graft_entry__ as g; g.build()'` at the "
                "repo root. There is no CPU fallback for this path.") from exc
    handle = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = handle.pygsd_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libpygsd_hip.so ABI version {got} != expected {ABI_VERSION}; rebuild it")
    _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().pygsd_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed: {msg}")


# torch.cuda.current_stream() and `with torch.cuda.device(...)` cost 9 and 8 microseconds of interpreter time each (device-index
# resolution through is_available / getenv on every call: profiles/r6b_host_profile.txt) -- a third of the host time of a step made
# of ~15 launches.  The C entry points behind them take a fraction of a microsecond.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """Raw hipStream_t of torch's current stream on the current device."""
    if _raw_stream is not None and _cur_device is not None:
        return c_void_p(_raw_stream(_cur_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


_STREAM_OBJECTS = {}


def current_stream():
    """torch.cuda.current_stream() without its per-call cost: the Stream object of a (device, raw stream) pair is looked up
    once (torch's streams live in a pool for the life of the process)."""
    if _raw_stream is None or _cur_device is None:
        return torch.cuda.current_stream()
    dev = _cur_device()
    key = (dev, _raw_stream(dev))
    s = _STREAM_OBJECTS.get(key)
    if s is None:
        s = _STREAM_OBJECTS[key] = torch.cuda.current_stream()
    return s


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(dev):
    """`with on_device(t.device):` -- torch.cuda.device(dev), but free when `dev` is the current device already (the
    one-process-per-GPU case: always)."""
    idx = getattr(dev, "index", dev)
    if _cur_device is not None and (idx is None or idx == _cur_device()):
        return _NO_GUARD
    return torch.cuda.device(dev)


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def fingerprint(t):
    """64-bit content fingerprint of a device tensor, as an int64[1] device tensor (pygsd_fingerprint_u64; queued, no host
    read).  Non-contiguous views are fingerprinted through a contiguous copy."""
    src = t.detach()
    if not src.is_contiguous():
        src = src.contiguous()
    out = torch.empty(1, dtype=torch.int64, device=src.device)
    with on_device(src.device):
        check(lib().pygsd_fingerprint_u64(c_void_p(src.data_ptr()), src.numel() * src.element_size(), c_void_p(out.data_ptr()),
                                          stream_ptr()), "pygsd_fingerprint_u64")
    return out


def require_gpu(*tensors):
    """The path only exists on the GPU: refuse CPU tensors instead of silently computing elsewhere."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "pytorch_geometric_signed_directed_amd ops run on MI355X (HIP) only; got a "
                f"{t.device} tensor. There is no CPU fallback (the CPU restatement lives in oracle/ "
                "and is test infrastructure).")


_I64_MAX, _I64_MIN = (1 << 63) - 1, -(1 << 63)


def check_node_ids(*bounded_lists, what="edge_index"):
    """check_node_ids((bound, ids), (bound, ids), ...): raise IndexError unless every id of every list lies in
    [0, its bound) -- where the reference's index_select / scatter_add_ raise.  One small kernel per list and
    ONE device->host read for all of them."""
    lists = [(int(b), t) for b, t in bounded_lists if t is not None and t.numel()]
    if not lists:
        return
    dev = lists[0][1].device
    minmax = torch.tensor([[_I64_MAX, _I64_MIN]] * len(lists), dtype=torch.int64, device=dev)
    with on_device(dev):
        for k, (_, t) in enumerate(lists):
            if t.dtype != torch.int64:
                raise TypeError(f"{what} must be int64 (torch.long), got {t.dtype}")
            t = t.contiguous()
            check(lib().pygsd_id_range_i64(ptr(t), t.numel(), c_void_p(minmax.data_ptr() + 16 * k), stream_ptr()),
                  "pygsd_id_range_i64")
    for (bound, _), (lo, hi) in zip(lists, minmax.tolist()):
        if lo < 0 or hi >= bound:
            raise IndexError(f"{what} holds node id {lo if lo < 0 else hi}, outside [0, {bound}); the HIP path "
                             "gathers and scatters rows by these ids")


# ---- library routes ---------------------------------------------------------------------------
# The path is hand-written HIP; where a shape falls outside what the kernels tile, the host may still route a DENSE
# product or reduction through a library (hipBLASLt / rocBLAS behind torch.mm, a torch reduction).  Every such route is
# counted here and announced once per (site, shape): "hand-written on the hot path" is an invariant the GPU tests assert
# (zero routes for the BASELINE configurations with default switches), not a claim.
_LIBRARY_ROUTES = collections.Counter()
_LIBRARY_WARNED = set()
_library_lock = threading.Lock()


def c16(t):
    """`t` contiguous AND starting on a 16-byte boundary, as the float4 / bf16x8 kernels address their operands.  `.contiguous()`
    alone does not give that: a ONE-ROW column slice of a wider matrix is contiguous as it stands, wherever it starts (a graph of
    one node fed from a padded feature matrix: found by tests/test_gpu_fuzz.py).  Copies only then."""
    if t is None:
        return None
    # (a one-row matrix also keeps whatever row stride its parent had -- meaningless, but the kernels are told it)
    ok = t.is_contiguous() and t.data_ptr() % 16 == 0 and (t.dim() != 2 or t.size(0) != 1 or t.stride(0) == t.size(1))
    return t if ok else t.clone(memory_format=torch.contiguous_format)


def note_library_route(site, detail=""):
    """Record one library-routed dense product / reduction at `site` (warns once per (site, detail))."""
    key = (site, str(detail))
    with _library_lock:
        _LIBRARY_ROUTES[site] += 1
        first = key not in _LIBRARY_WARNED
        _LIBRARY_WARNED.add(key)
    if first:
        warnings.warn(f"pytorch_geometric_signed_directed_amd: {site} runs through a library (not the HIP kernels): {detail}",
                      RuntimeWarning, stacklevel=3)


def library_routes():
    """{site: count} of the library-routed products / reductions since the last reset."""
    with _library_lock:
        return dict(_LIBRARY_ROUTES)


def reset_library_routes():
    with _library_lock:
        _LIBRARY_ROUTES.clear()


# ---- kernel-timing recorder (bench.py) --------------------------------------------------------
def prof_enable(on=True):
    check(lib().pygsd_prof_enable(1 if on else 0), "pygsd_prof_enable")


def prof_reset():
    check(lib().pygsd_prof_reset(), "pygsd_prof_reset")


def prof_collect(kernel):
    n, ms = c_int64(0), c_double(0.0)
    check(lib().pygsd_prof_collect(KERNEL_IDS[kernel], ctypes.byref(n), ctypes.byref(ms)),
          "pygsd_prof_collect")
    return n.value, ms.value
