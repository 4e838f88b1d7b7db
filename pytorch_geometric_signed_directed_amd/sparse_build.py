"""Device coalesce (sort by (row, col), merge duplicates by summation): the
torch_geometric.utils.coalesce step of the operator build
(reference utils/directed/get_magnetic_Laplacian.py:60)."""
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
    with torch.cuda.device(keys.device):
        need = ctypes.c_size_t(0)
        check(lib.pygsd_sort_keys_u64_workspace(n, ctypes.byref(need)), "pygsd_sort_keys_u64_workspace")
        ws = torch.empty(need.value, dtype=torch.uint8, device=keys.device)
        check(lib.pygsd_sort_keys_u64(ptr(keys), ptr(out), ptr(perm), n, int(key_bits), ptr(ws), need.value,
                                      stream_ptr()), "pygsd_sort_keys_u64")
    return out, perm


def coalesce_sum(index: Tensor, attr: Tensor, n: int) -> Tuple[Tensor, Tensor]:
    """index [2, M] int64, attr [M, C] -> (unique index sorted by (row, col), summed attr)."""
    m = index.size(1)
    if m == 0:
        return index, attr
    key = index[0] * n + index[1]
    bits = max(1, int(n * n - 1).bit_length()) if n > 1 else 1
    skey, perm = sort_keys(key, bits)
    perm = perm.long()
    head = torch.ones(m, dtype=torch.bool, device=index.device)
    head[1:] = skey[1:] != skey[:-1]
    seg = head.long().cumsum(0) - 1
    ukey = skey[head]
    out_index = torch.stack([ukey // n, ukey % n])
    sums = torch.zeros((ukey.numel(), attr.size(1)), dtype=attr.dtype, device=attr.device)
    sums.index_add_(0, seg, attr[perm])
    return out_index, sums
