"""Prob_Imbalance_Loss -- drop-in for torch_geometric_signed_directed/utils/directed/prob_imbalance_loss.py:6
(DIGRAC's probabilistic imbalance objective), SURVEY.md 8(f) rank 4.

The reference evaluates, for every ordered cluster pair, w_kl = P[:, k]^T A P[:, l] with one sparse
mat-vec each (K^2 of them) plus K more for the volumes.  Here the whole flow matrix W = P^T (A P) is ONE
HIP SpMM (A P, width K) and one [K, N] x [N, K] product; the volumes are (colsum(A) + rowsum(A)) . P.
The pair-selection logic on the K x K scalars (normalisation, 'sort' / 'std' / 'naive' thresholding, the
`.item()` branches) is restated unchanged -- including the reference's behaviour that only the 'sort'
branch keeps the autograd graph (the other branches rebuild a FloatTensor from the values)."""
from typing import Optional, Union

import numpy as np
import torch

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
        self._memo = None

    def _operator(self, A):
        """(pattern computing A @ X, values, colsum(A) + rowsum(A)) for a sparse COO / dense adjacency."""
        m = self._memo
        if m is not None and m[0] is A and m[1] == A._version:
            return m[2]
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
        self._memo = (A, A._version, out) if not A.is_sparse else None
        return out

    def forward(self, P: torch.FloatTensor, A: torch.Tensor, K: int, normalization: str = 'vol_sum',
                threshold: str = 'sort') -> torch.FloatTensor:
        assert normalization in ['vol_sum', 'vol_min', 'vol_max',
                                 'plain'], 'Please input the correct normalization method name!'
        assert threshold in ['sort', 'std', 'naive'], 'Please input the correct threshold method name!'
        device = A.device
        epsilon = torch.FloatTensor([1e-8]).to(device)
        pat, val, deg = self._operator(A)
        flow = torch.matmul(P[:, :K].t(), spmm(pat, P[:, :K].contiguous(), val))   # flow[k, l] = P_k^T A P_l
        vol = torch.matmul(deg, P[:, :K])
        second_max_vol = torch.topk(vol, 2).values[1] + epsilon
        result = torch.zeros(1).to(device)
        imbalance = []
        imbalance_std = []
        for k in range(K - 1):
            for l in range(k + 1, K):  # noqa: E741
                w_kl, w_lk = flow[k, l], flow[l, k]
                if (w_kl - w_lk).item() != 0:
                    if normalization == 'vol_sum':
                        curr = torch.abs(w_kl - w_lk) / (vol[k] + vol[l] + epsilon) * 2
                    elif normalization == 'vol_min':
                        curr = torch.abs(w_kl - w_lk) / (w_kl + w_lk) * torch.min(vol[k], vol[l]) / second_max_vol
                    elif normalization == 'vol_max':
                        curr = torch.abs(w_kl - w_lk) / (torch.max(vol[k], vol[l]) + epsilon)
                    else:
                        curr = torch.abs(w_kl - w_lk) / (w_kl + w_lk)
                    if threshold != 'std' or np.power((w_kl - w_lk).item(), 2) - 9 * (w_kl + w_lk).item() > 0:
                        imbalance.append(curr)
                    else:
                        imbalance_std.append(curr)
        imbalance_values = [curr.item() for curr in imbalance]
        if threshold == 'sort':
            ind_sorted = np.argsort(-np.array(imbalance_values))
            for ind in ind_sorted[:int(self.sel)]:
                result += imbalance[ind]
            return torch.ones(1, requires_grad=True).to(device) - result / self.sel
        elif len(imbalance) > 0:
            return torch.ones(1, requires_grad=True).to(device) - torch.mean(torch.FloatTensor(imbalance)).to(device)
        elif threshold == 'std':
            return torch.ones(1, requires_grad=True).to(device) - torch.mean(torch.FloatTensor(imbalance_std)).to(device)
        else:
            return torch.ones(1, requires_grad=True).to(device)
