"""Host wrapper of the stand-alone device key sort (pygsd_sort_keys_u64): a stable radix sort of 64-bit
keys with the permutation, the primitive under torch_geometric.utils.coalesce
(reference utils/directed/get_magnetic_Laplacian.py:60).  The Laplacian build fuses its own sort
(pygsd_maglap_sort); this entry point is kept for callers that coalesce other COO data."""
import ctypes
from typing import Tuple

import torch

from . import _cabi
from ._cabi import check, ptr, stream_ptr

Tensor = torch.Tensor


def sort_keys(keys: Tensor, key_bits: int) -> Tuple[Tensor, Tensor]:
    """Stable device radix sort of non-negative int64 keys -> (sorted keys, int32 permutation)."""
    _cabi.require_gpu(keys)
    keys = keys.contiguous()
    n = keys.numel()
    out = torch.empty_like(keys)
    perm = torch.empty(n, dtype=torch.int32, device=keys.device)
    if n == 0:
        return out, perm
    lib = _cabi.lib()
    with _cabi.on_device(keys.device):
        need = ctypes.c_size_t(0)
        check(lib.pygsd_sort_keys_u64_workspace(n, ctypes.byref(need)), "pygsd_sort_keys_u64_workspace")
        ws = torch.empty(need.value, dtype=torch.uint8, device=keys.device)
        check(lib.pygsd_sort_keys_u64(ptr(keys), ptr(out), ptr(perm), n, int(key_bits), ptr(ws), need.value,
                                      stream_ptr()), "pygsd_sort_keys_u64")
    return out, perm
