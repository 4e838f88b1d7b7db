"""ORACLE -- TEST INFRASTRUCTURE.  Records the block statistics of the REFERENCE's own SBM generators
(tests/golden/sbm_stats.npz).  Runs ONLY in the build container (needs /root/reference and networkx); the GPU box
never runs it.

    python oracle/gen_sbm_stats.py

The benchmark graphs (1M nodes) cannot come from the reference generators (networkx / Python loops, quadratic in
N), so bench.py and the tests draw them from the vectorised samplers of
pytorch_geometric_signed_directed_amd/graphs.py.  This script runs the reference's unmodified
`DSBM` (data/directed/DSBM.py:10-55), `SSBM` (data/signed/SSBM.py:9-140) and `SDSBM` (data/general/SDSBM.py:10-67)
at N <= 2000 over SEEDS seeds and stores, per generator, the mean and standard deviation over the seeds of
  * DSBM / SDSBM: the [K, K] number of edges from cluster a to cluster b (SDSBM: also of the NEGATIVE edges),
  * SSBM: stored entries (positive, negative) inside clusters and across clusters,
so that tests/test_graph_samplers.py can hold the samplers to them (3 sigma of the difference of the means).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "pyg_shim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

from torch_geometric_signed_directed.data.directed.DSBM import DSBM  # noqa: E402
from torch_geometric_signed_directed.data.general.SDSBM import SDSBM  # noqa: E402
from torch_geometric_signed_directed.data.signed.SSBM import SSBM  # noqa: E402
from torch_geometric_signed_directed.utils.directed.meta_graph_generation import meta_graph_generation  # noqa: E402

SEEDS = 24
K = 5
DSBM_CFG = dict(N=2000, p=0.02, size_ratio=1.5, eta=0.1)          # mean total degree 40, as the benchmark graphs
SDSBM_CFG = dict(N=2000, p=0.02, size_ratio=1.5, eta=0.1)
SSBM_CFG = dict(n=2000, p=0.01, eta=0.1, size_ratio=2.0)


def block_counts(a, labels, k):
    a = sp.coo_matrix(a)
    key = labels[a.row] * k + labels[a.col]
    total = np.bincount(key, minlength=k * k).reshape(k, k)
    neg = np.bincount(key[a.data < 0], minlength=k * k).reshape(k, k)
    return total, neg


def main():
    out = {"seeds": SEEDS, "k": K}
    f = meta_graph_generation("cyclic", K, DSBM_CFG["eta"], False, 0.5)
    tot = []
    for seed in range(SEEDS):
        np.random.seed(1000 + seed)
        a, labels = DSBM(DSBM_CFG["N"], K, DSBM_CFG["p"], f, DSBM_CFG["size_ratio"])
        tot.append(block_counts(a, labels, K)[0])
    out.update(dsbm_n=DSBM_CFG["N"], dsbm_p=DSBM_CFG["p"], dsbm_size_ratio=DSBM_CFG["size_ratio"],
               dsbm_eta=DSBM_CFG["eta"], dsbm_meta=f, dsbm_mean=np.mean(tot, 0), dsbm_std=np.std(tot, 0, ddof=1))

    fs = f.copy()
    for i in range(K):
        for j in range(K):
            if (i + j) % 2:
                fs[i, j] = -fs[i, j]                                # test/general_test.py:28-31
    tot, neg = [], []
    for seed in range(SEEDS):
        np.random.seed(2000 + seed)
        a, labels = SDSBM(SDSBM_CFG["N"], K, SDSBM_CFG["p"], fs, SDSBM_CFG["size_ratio"], SDSBM_CFG["eta"])
        t, m = block_counts(a, labels, K)
        tot.append(t)
        neg.append(m)
    out.update(sdsbm_n=SDSBM_CFG["N"], sdsbm_p=SDSBM_CFG["p"], sdsbm_size_ratio=SDSBM_CFG["size_ratio"],
               sdsbm_eta=SDSBM_CFG["eta"], sdsbm_meta=fs, sdsbm_mean=np.mean(tot, 0),
               sdsbm_std=np.std(tot, 0, ddof=1), sdsbm_neg_mean=np.mean(neg, 0), sdsbm_neg_std=np.std(neg, 0, ddof=1))

    rows = []
    for seed in range(SEEDS):
        np.random.seed(3000 + seed)
        (a_p, a_n), labels = SSBM(SSBM_CFG["n"], K, SSBM_CFG["p"], SSBM_CFG["eta"], size_ratio=SSBM_CFG["size_ratio"])
        stats = []
        for a in (a_p, a_n):
            a = sp.coo_matrix(a)
            same = labels[a.row] == labels[a.col]
            stats += [int(same.sum()), int((~same).sum())]
        rows.append(stats)                                           # [pos in, pos out, neg in, neg out] stored entries
    out.update(ssbm_n=SSBM_CFG["n"], ssbm_p=SSBM_CFG["p"], ssbm_eta=SSBM_CFG["eta"],
               ssbm_size_ratio=SSBM_CFG["size_ratio"], ssbm_mean=np.mean(rows, 0), ssbm_std=np.std(rows, 0, ddof=1))
    path = os.path.join(ROOT, "tests", "golden", "sbm_stats.npz")
    np.savez_compressed(path, **out)
    for key in ("dsbm_mean", "sdsbm_mean", "sdsbm_neg_mean", "ssbm_mean", "ssbm_std"):
        print(key, np.round(out[key], 1).tolist())
    print("wrote", path)


if __name__ == "__main__":
    main()
