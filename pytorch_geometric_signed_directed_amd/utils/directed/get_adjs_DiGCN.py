"""Operator pre-processing of DiGCN / DiGCL (reference utils/directed/get_adjs_DiGCN.py:9-254): the approximate
personalised-PageRank Laplacian and the second-order proximity matrix whose (edge_index, edge_weight) DiGCNConv
requires ("Normalized adj matrix cannot be None").  One-off host-side graph preparation, like in the reference
-- but without its dense N x N intermediates where the mathematics is sparse:

  * `get_second_directed_adj`: the reference materialises P, P^T P and P P^T as dense [N, N] tensors; here they
    are sparse products, so it scales with the number of 2-hop pairs instead of N^2.
  * `cal_fast_appr` / `fast_appr_power`: already sparse in the reference; restated.
  * `get_appr_directed_adj`: needs the dominant LEFT eigenvector of the (N+1) x (N+1) teleport-augmented
    transition matrix; the reference calls a dense eigen-solver.  Up to 2000 nodes the same solver is used (exact
    parity); beyond that a sparse power iteration finds the same Perron vector.
Results are returned on the device of `edge_index`, entries in row-major order like `torch.nonzero` yields them.
"""
from typing import Optional, Tuple, Union

import numpy as np
import scipy
import scipy.linalg
import scipy.sparse as sp
import torch


def _with_self_loops(edge_index, edge_weight, num_nodes, dtype):
    """add_self_loops(fill 1) as the reference calls it: existing loops are KEPT and n more appended."""
    ei = edge_index.detach().cpu().numpy()
    w = (np.ones(ei.shape[1]) if edge_weight is None else edge_weight.detach().cpu().double().numpy())
    loops = np.arange(num_nodes)
    return (np.concatenate([ei[0], loops]), np.concatenate([ei[1], loops]), np.concatenate([w, np.ones(num_nodes)]))


def _transition(edge_index, edge_weight, num_nodes, dtype):
    """Row-stochastic P = D^-1 (A + I) as a CSR matrix (duplicate entries add, as to_dense() adds them)."""
    r, c, w = _with_self_loops(edge_index, edge_weight, num_nodes, dtype)
    deg = np.zeros(num_nodes)
    np.add.at(deg, r, w)
    inv = np.where(deg != 0, 1.0 / np.where(deg != 0, deg, 1.0), 0.0)
    return sp.csr_matrix((inv[r] * w, (r, c)), shape=(num_nodes, num_nodes))


def _sym_normalised(L, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Non-zero entries of L in row-major order, scaled D^-1/2 L D^-1/2 with D = row sums."""
    L = L.tocsr()
    L.sum_duplicates()
    L.eliminate_zeros()
    L.sort_indices()
    coo = L.tocoo()
    deg = np.asarray(L.sum(axis=1)).reshape(-1)
    with np.errstate(divide="ignore"):
        dis = np.power(deg, -0.5)
    dis[np.isinf(dis)] = 0
    val = dis[coo.row] * coo.data * dis[coo.col]
    index = torch.from_numpy(np.stack([coo.row, coo.col]).astype(np.int64)).to(device)
    return index, torch.from_numpy(val.astype(np.float32)).to(device)


def fast_appr_power(A, alpha=0.1, max_iter=100, tol=1e-06, personalize=None):
    """Power iteration for the teleporting random walk of DiGCL (get_adjs_DiGCN.py:9-60); returns the
    symmetrised PageRank Laplacian (scipy sparse) and the stationary vector."""
    n, _ = A.shape
    r = np.asarray(A.sum(axis=1)).reshape(-1)
    k = r.nonzero()[0]
    D_1 = sp.csr_matrix((1 / r[k], (k, k)), shape=(n, n))
    if personalize is None:
        personalize = np.ones(n)
    personalize = personalize.reshape(n, 1)
    s = 1 / (1 + alpha) / n * personalize
    z_T = ((alpha * (1 + alpha)) * (r != 0) + ((1 - alpha) / (1 + alpha) + alpha * (1 + alpha)) * (r == 0))[np.newaxis, :]
    W = (1 - alpha) * A.T @ D_1
    x = s
    oldx = np.zeros((n, 1))
    iteration = 0
    while scipy.linalg.norm(x - oldx) > tol:
        oldx = x
        x = W @ x + s @ (z_T @ x)
        iteration += 1
        if iteration >= max_iter:
            break
    x = x / sum(x)
    x = x.reshape(-1)
    p = D_1 * A
    pi_sqrt = sp.diags(np.power(x, 0.5))
    pi_inv_sqrt = sp.diags(np.power(x, -0.5))
    L = (pi_sqrt * p * pi_inv_sqrt + pi_inv_sqrt * p.T * pi_sqrt) / 2.0
    L.data[np.isnan(L.data)] = 0.0
    return L, x


def cal_fast_appr(alpha: float, edge_index: torch.LongTensor, num_nodes: Union[int, None], dtype: torch.dtype,
                  edge_weight: Optional[torch.FloatTensor] = None) -> Tuple[torch.LongTensor, torch.FloatTensor]:
    r, c, w = _with_self_loops(edge_index, edge_weight, num_nodes, dtype)
    adj = sp.csr_matrix((w.astype(np.float32), (r, c)), shape=(num_nodes, num_nodes))
    L, _ = fast_appr_power(adj, alpha=alpha, tol=1e-6)
    L = L.tocoo()
    index = torch.from_numpy(np.vstack((L.row, L.col)).astype(np.int64)).to(edge_index.device)
    values = torch.from_numpy(np.asarray(L.data, dtype=np.float32)).to(edge_index.device)
    deg = torch.zeros(num_nodes, dtype=values.dtype, device=values.device).index_add_(0, index[0], values)
    dis = deg.pow(-0.5)
    dis[dis == float('inf')] = 0
    return index, dis[index[0]] * values * dis[index[1]]


def _perron_left_vector(p, alpha: float, n: int, dense_limit: int = 2000) -> np.ndarray:
    """Dominant left eigenvector (first n components) of [[(1-alpha) P, alpha 1], [1^T / n, 0]]."""
    if n <= dense_limit:                             # the reference's dense solver: exact parity
        pv = np.zeros((n + 1, n + 1), dtype=np.float32)
        pv[:n, :n] = (1 - alpha) * p.toarray()
        pv[n, :n] = 1.0 / n
        pv[:n, n] = alpha
        eig_value, left_vector = scipy.linalg.eig(pv, left=True, right=False)
        return left_vector.real[:, np.argsort(-eig_value.real, kind="stable")[0]][:n].astype(np.float64)
    pt = ((1 - alpha) * p).T.tocsr()
    x, t = np.full(n, 1.0 / (n + 1)), 1.0 / (n + 1)
    for _ in range(1000):                            # x^T <- x^T M for the row-stochastic augmented M
        nx = pt @ x + t / n
        nt = alpha * x.sum()
        s = nx.sum() + nt
        nx, nt = nx / s, nt / s
        done = np.abs(nx - x).sum() + abs(nt - t) < 1e-12
        x, t = nx, nt
        if done:
            break
    return x


def get_appr_directed_adj(alpha: float, edge_index: torch.LongTensor, num_nodes: Union[int, None], dtype: torch.dtype,
                          edge_weight: Optional[torch.FloatTensor] = None) -> Tuple[torch.LongTensor, torch.FloatTensor]:
    """Approximate-PageRank Laplacian of DiGCN: L = (Pi^1/2 P Pi^-1/2 + Pi^-1/2 P^T Pi^1/2) / 2, sym-normalised."""
    p = _transition(edge_index, edge_weight, num_nodes, dtype)
    pi = _perron_left_vector(p, alpha, num_nodes)
    pi = pi / pi.sum()
    assert not (pi < 0).any()
    with np.errstate(divide="ignore"):
        pi_inv_sqrt = np.power(pi, -0.5)
    pi_inv_sqrt[np.isinf(pi_inv_sqrt)] = 0
    pi_sqrt = np.power(pi, 0.5)
    half = sp.diags(pi_sqrt) @ p @ sp.diags(pi_inv_sqrt)
    return _sym_normalised((half + half.T) / 2.0, edge_index.device)


def get_second_directed_adj(edge_index: torch.LongTensor, num_nodes: Union[int, None], dtype: torch.dtype,
                            edge_weight: Optional[torch.FloatTensor] = None) -> Tuple[torch.LongTensor, torch.FloatTensor]:
    """Second-order proximity of DiGCN: L = (L_in * [L_out != 0] + L_out * [L_in != 0]) / 2 with L_in = P^T P,
    L_out = P P^T, sym-normalised."""
    p = _transition(edge_index, edge_weight, num_nodes, dtype)
    l_in, l_out = (p.T @ p).tocsr(), (p @ p.T).tocsr()
    l_in.eliminate_zeros()
    l_out.eliminate_zeros()
    both = l_in.multiply(l_out != 0) + l_out.multiply(l_in != 0)
    return _sym_normalised(both / 2.0, edge_index.device)
