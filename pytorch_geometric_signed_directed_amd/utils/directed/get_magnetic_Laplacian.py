"""get_magnetic_Laplacian -- drop-in for
torch_geometric_signed_directed/utils/directed/get_magnetic_Laplacian.py:10, computed on the GPU."""
from typing import Optional

import torch

from .._laplacian import laplacian_parts, laplacian_values


def _assemble(edge_index, edge_weight, normalization, dtype, num_nodes, q, return_lambda_max, signed,
              absolute_degree):
    if normalization is not None:
        assert normalization in ['sym'], 'Invalid normalization'
    if num_nodes is None:
        keep = edge_index[0] != edge_index[1]
        rest = edge_index[:, keep]
        num_nodes = int(rest.max()) + 1 if rest.numel() > 0 else 0
    parts = laplacian_parts(edge_index, edge_weight, num_nodes, signed, absolute_degree, dtype)
    off_r, off_i, diag = laplacian_values(parts, q, normalization)
    loops = torch.arange(num_nodes, dtype=torch.long, device=edge_index.device).unsqueeze(0).repeat(2, 1)
    index = torch.cat([parts.index, loops], dim=1)
    real = torch.cat([off_r, diag])
    imag = torch.cat([off_i, torch.zeros_like(diag)])
    if not return_lambda_max:
        return index, real, imag
    import numpy as np
    import scipy.sparse as sp
    from scipy.sparse.linalg import eigsh
    val = torch.complex(real, imag).detach().cpu().numpy()
    idx = index.cpu().numpy()
    L = sp.coo_matrix((val, (idx[0], idx[1])), (num_nodes, num_nodes))
    lam = eigsh(L, k=1, which='LM', return_eigenvectors=False)
    return index, real, imag, float(np.asarray(lam).real.item())


def get_magnetic_Laplacian(edge_index: torch.LongTensor, edge_weight: Optional[torch.Tensor] = None,
                           normalization: Optional[str] = 'sym', dtype: Optional[int] = None,
                           num_nodes: Optional[int] = None, q: Optional[float] = 0.25,
                           return_lambda_max: bool = False):
    r"""Magnetic Laplacian of the digraph: returns (edge_index, real, imag[, lambda_max]) with the
    E_s coalesced off-diagonal entries sorted by (row, col) followed by N self loops, exactly the
    reference's output layout."""
    return _assemble(edge_index, edge_weight, normalization, dtype, num_nodes, q, return_lambda_max,
                     False, True)
