"""Prob_Imbalance_Loss -- drop-in for torch_geometric_signed_directed/utils/directed/prob_imbalance_loss.py:6
(DIGRAC's probabilistic imbalance objective), SURVEY.md 8(f) rank 4.

The reference evaluates, for every ordered cluster pair, w_kl = P[:, k]^T A P[:, l] with one sparse
mat-vec each (K^2 of them) plus K more for the volumes.  Here the whole flow matrix W = P^T (A P) is ONE
HIP SpMM (A P, width K) and one [K, N] x [N, K] product; the volumes are (colsum(A) + rowsum(A)) . P.
The pair selection on the K (K - 1) / 2 scalars (normalisation, 'sort' / 'std' / 'naive' thresholding) is
evaluated for all pairs at once on the device instead of pair by pair with host reads -- keeping the reference's
behaviour that only the 'sort' branch carries the autograd graph (the others rebuild a FloatTensor from values)."""
from typing import Optional, Union

import numpy as np
import torch

from ...dense import matmul
from ...memo import TensorMemo
from ...sparse import Pattern, spmm


class Prob_Imbalance_Loss(torch.nn.Module):
    def __init__(self, F: Optional[Union[int, np.ndarray]] = None):
        super().__init__()
        if isinstance(F, int):
            self.sel = F
        elif F is not None:
            K = F.shape[0]
            self.sel = 0
            for i in range(K - 1):
                for j in range(i + 1, K):
                    if (F[i, j] + F[j, i]) > 0:
                        self.sel += 1
        self._memo = TensorMemo(1)

    def _operator(self, A):
        """(pattern computing A @ X, values, colsum(A) + rowsum(A)) for a sparse COO / dense adjacency."""
        hit = None if A.is_sparse else self._memo.get((A,), "operator")
        if hit is not None:
            return hit
        if A.is_sparse:
            A = A.coalesce()
            idx, val = A.indices(), A.values().float()
        else:
            idx = A.nonzero(as_tuple=False).t().contiguous()
            val = A[idx[0], idx[1]].float()
        n = A.size(0)
        # (A X)[row] = sum_col A[row, col] X[col]: gather at col, scatter at row
        pat = Pattern(torch.stack([idx[1], idx[0]]), n, n)
        deg = torch.zeros(n, dtype=torch.float32, device=val.device)
        deg = deg.index_add(0, idx[0], val).index_add(0, idx[1], val)
        out = (pat, val, deg)
        # (dense adjacency: one nonzero() scan per tensor and in-place version; memo.TensorMemo -- weakly held, its opt-outs
        #  and strict mode apply.  A sparse tensor has no version counter worth trusting: rebuilt per call)
        return out if A.is_sparse else self._memo.put((A,), "operator", out)

    def forward(self, P: torch.FloatTensor, A: torch.Tensor, K: int, normalization: str = 'vol_sum',
                threshold: str = 'sort') -> torch.FloatTensor:
        assert normalization in ['vol_sum', 'vol_min', 'vol_max',
                                 'plain'], 'Please input the correct normalization method name!'
        assert threshold in ['sort', 'std', 'naive'], 'Please input the correct threshold method name!'
        device = A.device
        eps = 1e-8
        pat, val, deg = self._operator(A)
        prob = P[:, :K]
        flow = matmul(prob.t(), spmm(pat, prob.contiguous(), val))                # flow[k, l] = P_k^T A P_l
        vol = matmul(deg.unsqueeze(0), prob).squeeze(0)                           # probabilistic cluster volumes
        # all cluster pairs k < l at once (the reference loops over them with a host read per pair)
        k_idx, l_idx = torch.triu_indices(K, K, offset=1, device=device)
        forth, back = flow[k_idx, l_idx], flow[l_idx, k_idx]
        gap, total = forth - back, forth + back
        if normalization == 'vol_sum':
            score = gap.abs() / (vol[k_idx] + vol[l_idx] + eps) * 2
        elif normalization == 'vol_min':
            score = gap.abs() / total * torch.minimum(vol[k_idx], vol[l_idx]) / (torch.topk(vol, 2).values[1] + eps)
        elif normalization == 'vol_max':
            score = gap.abs() / (torch.maximum(vol[k_idx], vol[l_idx]) + eps)
        else:
            score = gap.abs() / total
        live = gap != 0                                                           # pairs with any imbalance at all
        one = torch.ones(1, requires_grad=True).to(device)
        if threshold == 'sort':
            # the `sel` largest scores, summed WITH their autograd graph (the only differentiable branch, as in
            # the reference); fewer live pairs than `sel` just contribute fewer terms to the same divisor
            picked = torch.argsort(score.detach().masked_fill(~live, float('-inf')), descending=True)[:int(self.sel)]
            picked = picked[live[picked]]
            return one - score[picked].sum().reshape(1) / self.sel
        significant = live & (gap.pow(2) - 9 * total > 0) if threshold == 'std' else live
        values = score.detach()                                                   # value only: the reference rebuilds a
        if bool(significant.any()):                                               # FloatTensor from the numbers here
            return one - values[significant].mean()
        if threshold == 'std' and bool(live.any()):
            return one - values[live].mean()
        if threshold == 'std':
            return one - torch.mean(torch.empty(0, device=device))                # reference: mean of an empty list -> nan
        return one
