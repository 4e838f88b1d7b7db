"""Initial embeddings of SGCN / SNEA (utils/signed/create_spectral_features.py:8-41): truncated SVD of the
symmetrised signed adjacency.  A one-off CPU preprocessing step in the reference too (it moves the edges to the
CPU and calls scikit-learn); not part of the device path."""
import numpy as np
import scipy.sparse as sp
import torch
from sklearn.decomposition import TruncatedSVD


def create_spectral_features(pos_edge_index: torch.Tensor, neg_edge_index: torch.Tensor, node_num: int,
                             dim: int) -> torch.Tensor:
    pos = pos_edge_index.detach().cpu().numpy()
    neg = neg_edge_index.detach().cpu().numpy()
    row = np.concatenate([pos[0], neg[0]])
    col = np.concatenate([pos[1], neg[1]])
    val = np.concatenate([np.full(pos.shape[1], 2.0, np.float32), np.zeros(neg.shape[1], np.float32)])
    # both orientations; duplicates add (coalesce) BEFORE the shift, listed pairs only: +1 / -1 for a single
    # positive / negative listing
    a = sp.coo_matrix((np.concatenate([val, val]), (np.concatenate([row, col]), np.concatenate([col, row]))),
                      shape=(node_num, node_num)).tocsr()
    a.sum_duplicates()
    a.data = a.data - 1.0
    svd = TruncatedSVD(n_components=dim, n_iter=128)
    svd.fit(a)
    return torch.from_numpy(svd.components_.T.copy()).to(torch.float)
