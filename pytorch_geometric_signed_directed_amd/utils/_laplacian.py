"""Device-side build of the (signed) magnetic Laplacian operator -- SURVEY.md 8(a) rows a3/a4.

Follows utils/directed/get_magnetic_Laplacian.py:10-93 and
utils/general/get_magnetic_signed_Laplacian.py:10-98 of the reference: drop self loops ->
symmetrise -> coalesce(add) of [w, +-w(, |w|)] -> A_s = sum/2, Theta = 2 pi q (w_uv - w_vu) ->
degree -> D^-1/2 A_s D^-1/2 (.) exp(i Theta) -> L = I - H (sym) or D - A_s (.) exp(i Theta) (None).

All arrays stay on the GPU.  Two host round-trips are inherent (the number of surviving
non-loop entries and the number of distinct symmetrised entries size the outputs), exactly as in
the reference's boolean-mask / coalesce calls.
"""
import math
from typing import Optional, Tuple

import torch

from .. import _cabi
from ..sparse_build import coalesce_sum

Tensor = torch.Tensor


class LaplacianParts:
    """Symmetrised pattern + the ingredients of the operator values (all COO, sorted by (row, col))."""
    __slots__ = ("index", "a_sym", "theta", "deg", "n")

    def __init__(self, index, a_sym, theta, deg, n):
        self.index, self.a_sym, self.theta, self.deg, self.n = index, a_sym, theta, deg, n


def laplacian_parts(edge_index: Tensor, edge_weight: Optional[Tensor], n: int, signed: bool,
                    absolute_degree: bool, dtype) -> LaplacianParts:
    _cabi.require_gpu(edge_index, edge_weight)
    row, col = edge_index[0], edge_index[1]
    keep = row != col
    row, col = row[keep], col[keep]
    if edge_weight is None:
        w = torch.ones(row.numel(), dtype=dtype or torch.float32, device=edge_index.device)
    else:
        w = edge_weight[keep]
    both = torch.stack([torch.cat([row, col]), torch.cat([col, row])])
    cols = [torch.cat([w, w]), torch.cat([w, -w])]
    if signed:
        cols.append(torch.cat([w.abs(), w.abs()]))
    index, sums = coalesce_sum(both, torch.stack(cols, dim=1), n)
    a_sym = sums[:, 0] / 2
    if not signed:
        deg_src = a_sym
    elif absolute_degree:
        deg_src = sums[:, 2] / 2
    else:
        deg_src = a_sym.abs()
    deg = torch.zeros(n, dtype=a_sym.dtype, device=a_sym.device).index_add_(0, index[0], deg_src)
    return LaplacianParts(index, a_sym, sums[:, 1], deg, n)


def laplacian_values(parts: LaplacianParts, q, normalization: Optional[str]) -> Tuple[Tensor, Tensor, Tensor]:
    """(off_real, off_imag, diag) of L; differentiable in q when q is a tensor that requires grad."""
    row, col = parts.index[0], parts.index[1]
    phase_arg = (2 * math.pi * q) * parts.theta
    cos, sin = torch.cos(phase_arg), torch.sin(phase_arg)
    if normalization is None:
        mag = parts.a_sym
        diag = parts.deg
    else:
        dis = parts.deg.pow(-0.5)
        dis = dis.masked_fill(dis == float("inf"), 0)
        mag = dis[row] * parts.a_sym * dis[col]
        diag = torch.ones_like(parts.deg)
    return -(mag * cos), -(mag * sin), diag
