"""get_magnetic_signed_Laplacian -- drop-in for
torch_geometric_signed_directed/utils/general/get_magnetic_signed_Laplacian.py:10 (GPU)."""
from typing import Optional

import torch

from ..directed.get_magnetic_Laplacian import _assemble


def get_magnetic_signed_Laplacian(edge_index: torch.LongTensor, edge_weight: Optional[torch.Tensor] = None,
                                  normalization: Optional[str] = 'sym', dtype: Optional[int] = None,
                                  num_nodes: Optional[int] = None, q: Optional[float] = 0.25,
                                  return_lambda_max: bool = False, absolute_degree: bool = True):
    r"""Signed magnetic Laplacian (degree from |w| when absolute_degree, else from |A_s|)."""
    return _assemble(edge_index, edge_weight, normalization, dtype, num_nodes, q, return_lambda_max,
                     True, absolute_degree)
