"""Synthetic benchmark inputs: vectorised samplers for the reference's stochastic block models
(distributions of data/directed/DSBM.py:10-55 + utils/directed/meta_graph_generation.py:6-94,
data/signed/SSBM.py:9-140 and data/general/SDSBM.py:10-67).  tests/test_graph_samplers.py holds their
block-pair edge counts and sign fractions to the statistics of the reference generators themselves
(tests/golden/sbm_stats.npz, recorded by oracle/gen_sbm_stats.py).  The reference generators go through networkx / Python loops and are
quadratic in N (19 s at N = 20k), so they cannot produce the 1M-node benchmark
graphs; these samplers draw from the same block-pair edge distributions in O(E).

Host-side numpy; used by bench.py and tests only (not part of the layer path).
"""
import math
from typing import Tuple

import numpy as np


def cyclic_meta_graph(k: int = 5, eta: float = 0.1, fill_val: float = 0.5) -> np.ndarray:
    """meta_graph_generation('cyclic', K, eta, ambient=False, fill_val) for K > 2:
    diagonal 0.5, forward 1-eta, backward eta, all other pairs fill_val."""
    if k <= 2:
        raise ValueError("cyclic meta-graph sampler expects K > 2")
    f = np.full((k, k), float(fill_val))
    np.fill_diagonal(f, 0.5)
    for i in range(k):
        j = (i + 1) % k
        f[i, j] = 1.0 - eta
        f[j, i] = eta
    return f


def block_sizes(n: int, k: int, size_ratio: float) -> np.ndarray:
    """Geometric cluster sizes (DSBM.py:33-44): largest = size_ratio x smallest."""
    if size_ratio > 1:
        r = size_ratio ** (1.0 / (k - 1))
        sizes = [math.floor(n * (1 - r) / (1 - r ** k))]
        for _ in range(1, k - 1):
            sizes.append(math.floor(sizes[-1] * r))
        sizes.append(n - sum(sizes))
    else:
        sizes = [math.floor((i + 1) * n / k) - math.floor(i * n / k) for i in range(k)]
    return np.asarray(sizes, dtype=np.int64)


def _sample_block(rng, n_rows, n_cols, prob, same_block):
    """Bernoulli(prob) over the ordered pairs of one block pair, sampled as a Binomial count of
    uniformly drawn pairs (duplicates removed; for sparse p the bias is O(p))."""
    pairs = n_rows * n_cols - (n_rows if same_block else 0)
    if pairs <= 0 or prob <= 0:
        return np.empty(0, np.int64), np.empty(0, np.int64)
    m = rng.binomial(pairs, min(prob, 1.0))
    if m == 0:
        return np.empty(0, np.int64), np.empty(0, np.int64)
    r = rng.integers(0, n_rows, m, dtype=np.int64)
    c = rng.integers(0, n_cols, m, dtype=np.int64)
    if same_block:
        keep = r != c
        r, c = r[keep], c[keep]
    key = np.unique(r * n_cols + c)
    return key // n_cols, key % n_cols


def dsbm(n: int, k: int, p: float, meta: np.ndarray, size_ratio: float = 1.5, seed: int = 0
         ) -> Tuple[np.ndarray, np.ndarray]:
    """Directed SBM: edge u -> v (u != v) with probability p * meta[c(u), c(v)], node ids permuted.
    Returns (edge_index int64 [2, E], labels int64 [N])."""
    rng = np.random.default_rng(seed)
    sizes = block_sizes(n, k, size_ratio)
    starts = np.concatenate([[0], np.cumsum(sizes)])
    rows, cols = [], []
    for a in range(k):
        for b in range(k):
            r, c = _sample_block(rng, int(sizes[a]), int(sizes[b]), p * meta[a, b], a == b)
            rows.append(r + starts[a])
            cols.append(c + starts[b])
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    relabel = rng.permutation(n)
    labels = np.empty(n, dtype=np.int64)
    labels[relabel] = np.repeat(np.arange(k), sizes)
    ei = np.stack([relabel[rows], relabel[cols]])
    order = rng.permutation(ei.shape[1])  # COO order carries no structure
    return np.ascontiguousarray(ei[:, order]), labels                  # (fancy indexing yields F-order: rows strided)


def dsbm_for_edges(n: int, e_target: int, k: int = 5, eta: float = 0.1, size_ratio: float = 1.5,
                   seed: int = 0):
    """DSBM with the cyclic meta-graph and p chosen so that E[#edges] = e_target
    (p = 4.0e-4 for 100k / 2M, 4.0e-5 for 1M / 20M)."""
    meta = cyclic_meta_graph(k, eta, 0.5)
    sizes = block_sizes(n, k, size_ratio).astype(np.float64)
    pairs = np.outer(sizes, sizes)
    pairs[np.diag_indices(k)] -= sizes
    p = e_target / float((pairs * meta).sum())
    ei, labels = dsbm(n, k, p, meta, size_ratio, seed)
    return ei, labels, p


def ssbm(n: int, k: int, p: float, eta: float, size_ratio: float = 2.0, seed: int = 0
         ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Signed SBM (SSBM.py with pin = pout = p, etain = etaout = eta, values='ones'): each unordered
    pair is an edge w.p. p; sign +1 inside a cluster, -1 across, flipped w.p. eta; both orientations
    stored.  Returns (edge_index [2, 2M], sign float32 [2M], labels)."""
    rng = np.random.default_rng(seed)
    sizes = block_sizes(n, k, size_ratio)
    labels_sorted = np.repeat(np.arange(k), sizes)
    pairs = n * (n - 1) // 2
    m = rng.binomial(pairs, p)
    u = rng.integers(0, n, m, dtype=np.int64)
    v = rng.integers(0, n, m, dtype=np.int64)
    keep = u != v
    u, v = u[keep], v[keep]
    lo, hi = np.minimum(u, v), np.maximum(u, v)
    key = np.unique(lo * n + hi)
    lo, hi = key // n, key % n
    sign = np.where(labels_sorted[lo] == labels_sorted[hi], 1.0, -1.0).astype(np.float32)
    flip = rng.random(sign.size) < eta
    sign[flip] *= -1.0
    relabel = rng.permutation(n)
    labels = np.empty(n, dtype=np.int64)
    labels[relabel] = labels_sorted
    ei = np.stack([np.concatenate([relabel[lo], relabel[hi]]), np.concatenate([relabel[hi], relabel[lo]])])
    sign = np.concatenate([sign, sign])
    order = rng.permutation(ei.shape[1])
    return np.ascontiguousarray(ei[:, order]), sign[order], labels


def signed_cyclic_meta_graph(k: int = 5, eta: float = 0.1, fill_val: float = 0.5) -> np.ndarray:
    """The signed meta-graph of the reference's MSGNN tests (test/general_test.py:28-31): the cyclic DSBM
    meta-graph with F[i, j] negated where (i + j) is odd."""
    f = cyclic_meta_graph(k, eta, fill_val)
    i, j = np.indices(f.shape)
    f[(i + j) % 2 == 1] *= -1.0
    return f


def sdsbm(n: int, k: int, p: float, meta: np.ndarray, size_ratio: float = 1.5, eta: float = 0.1, seed: int = 0
          ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Signed directed SBM (data/general/SDSBM.py:10-67): a DSBM on |meta|; an edge between clusters (a, b) is
    negative where meta[a, b] < 0; then exactly floor(E * eta) edges chosen uniformly have their sign flipped.
    Returns (edge_index int64 [2, E], weight float32 [E] in {+1, -1}, labels int64 [N])."""
    ei, labels = dsbm(n, k, p, np.abs(meta), size_ratio, seed)
    sign = np.where(meta[labels[ei[0]], labels[ei[1]]] < 0, -1.0, 1.0).astype(np.float32)
    rng = np.random.default_rng([seed, 1])
    flip = rng.choice(sign.size, size=int(sign.size * eta), replace=False)
    sign[flip] *= -1.0
    return ei, sign, labels


def sdsbm_for_edges(n: int, e_target: int, k: int = 5, eta: float = 0.1, size_ratio: float = 1.5, seed: int = 0):
    """SDSBM on the signed cyclic meta-graph with p chosen so that E[#edges] = e_target (BASELINE config C4)."""
    meta = signed_cyclic_meta_graph(k, eta, 0.5)
    sizes = block_sizes(n, k, size_ratio).astype(np.float64)
    pairs = np.outer(sizes, sizes)
    pairs[np.diag_indices(k)] -= sizes
    p = e_target / float((pairs * np.abs(meta)).sum())
    ei, sign, labels = sdsbm(n, k, p, meta, size_ratio, eta, seed)
    return ei, sign, labels, p


def block_counts(edge_index: np.ndarray, labels: np.ndarray, k: int, weight=None) -> np.ndarray:
    """[k, k] number of edges from cluster a to cluster b (with `weight`: the sum of the weights instead)."""
    key = labels[edge_index[0]] * k + labels[edge_index[1]]
    return np.bincount(key, weights=weight, minlength=k * k).reshape(k, k)
