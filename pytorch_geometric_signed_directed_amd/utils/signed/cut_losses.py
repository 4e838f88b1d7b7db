"""SSSNET's probabilistic cut objectives on the HIP SpMM -- SURVEY.md 8(f) rank 4.

Drop-ins for utils/signed/prob_balanced_normalized_loss.py:8-48, prob_balanced_ratio_loss.py:8-43 and
unhappy_ratio.py:8-40 of the reference.  All three evaluate sum_k p_k^T M p_k (M = D_p - (A_p - A_n)) with
one sparse mat-vec PER CLUSTER through torch.sparse; here it is ONE SpMM Y = M P over all K columns
(pygsd_spmm_csr_f32) followed by column-wise dot products.  The constructors take the same scipy sparse
matrices and do the same one-time host preprocessing; the operator is uploaded on first use."""
import numpy as np
import scipy.sparse as sp
import torch

from ...sparse import Pattern, spmm


class _CutObjective(torch.nn.Module):
    def __init__(self, A_p: sp.spmatrix, A_n: sp.spmatrix):
        super().__init__()
        D_p = sp.diags(A_p.transpose().sum(axis=0).tolist(), [0]).tocsc()
        self._D_p = D_p
        mat = sp.coo_matrix(D_p - (A_p - A_n))
        self._n = mat.shape[0]
        self._rows = torch.from_numpy(mat.row.astype(np.int64))
        self._cols = torch.from_numpy(mat.col.astype(np.int64))
        self._vals = torch.from_numpy(mat.data.astype(np.float32))
        self._dev_cache = None

    def _operator(self, device):
        if self._dev_cache is None or self._dev_cache[0] != device:
            # Y[row] = sum_col M[row, col] P[col]: gather at col, scatter at row
            ei = torch.stack([self._cols, self._rows]).to(device)
            self._dev_cache = (device, Pattern(ei, self._n, self._n), self._vals.to(device))
        return self._dev_cache[1], self._dev_cache[2]

    def _quadratic_forms(self, prob: torch.Tensor) -> torch.Tensor:
        """[K] vector of p_k^T M p_k."""
        pat, vals = self._operator(prob.device)
        return (prob * spmm(pat, prob, vals)).sum(dim=0)


class Prob_Balanced_Normalized_Loss(_CutObjective):
    r"""Probabilistic balanced normalized cut loss of SSSNET: sum_k p_k^T M p_k / (p_k^T D_bar p_k + 1e-6)."""

    def __init__(self, A_p: sp.spmatrix, A_n: sp.spmatrix):
        super().__init__(A_p, A_n)
        D_n = sp.diags(A_n.transpose().sum(axis=0).tolist(), [0]).tocsc()
        self._d_bar = torch.from_numpy(np.asarray((self._D_p + D_n).diagonal(), dtype=np.float32))

    def forward(self, prob: torch.FloatTensor) -> torch.Tensor:
        d_bar = self._d_bar.to(prob.device)
        denominator = (prob * prob * d_bar.unsqueeze(1)).sum(dim=0) + 1e-6
        return (self._quadratic_forms(prob) / denominator).sum().reshape(1)


class Prob_Balanced_Ratio_Loss(_CutObjective):
    r"""Probabilistic balanced ratio cut loss of SSSNET: sum_k p_k^T M p_k / (p_k^T p_k + 1)."""

    def forward(self, prob: torch.FloatTensor) -> torch.Tensor:
        denominator = (prob * prob).sum(dim=0) + 1
        return (self._quadratic_forms(prob) / denominator).sum().reshape(1)


class Unhappy_Ratio(_CutObjective):
    r"""Ratio of unhappy edges: sum_k p_k^T M p_k / #edges."""

    def __init__(self, A_p: sp.spmatrix, A_n: sp.spmatrix):
        super().__init__(A_p, A_n)
        self.num_edges = len((A_p - A_n).nonzero()[0])

    def forward(self, prob: torch.FloatTensor) -> torch.Tensor:
        return (self._quadratic_forms(prob).sum() / self.num_edges).reshape(1)
