"""CPU: the vectorised SBM samplers that produce the benchmark / test graphs
(pytorch_geometric_signed_directed_amd/graphs.py) against the statistics of the REFERENCE generators, recorded in
this container by oracle/gen_sbm_stats.py from the reference's unmodified DSBM / SSBM / SDSBM at N = 2000 over 24
seeds (tests/golden/sbm_stats.npz).  For every block pair the mean count over the sampler's seeds must lie within
3 sigma of the reference's mean (sigma = standard error of the difference of the two means); block sizes are
checked exactly."""
import numpy as np

from conftest import load_golden
from pytorch_geometric_signed_directed_amd import graphs

SEEDS = 24


def _z(mean_a, std_a, n_a, mean_b, std_b, n_b):
    se = np.sqrt(std_a ** 2 / n_a + std_b ** 2 / n_b)
    return np.abs(mean_a - mean_b) / np.maximum(se, 1e-9)


def test_block_sizes_match_the_reference_geometric_sequence():
    # DSBM.py:33-44 at the survey's benchmark sizes
    assert graphs.block_sizes(100000, 5, 1.5).tolist() == [16163, 17887, 19795, 21906, 24249]
    assert graphs.block_sizes(1000000, 5, 1.5).tolist() == [161633, 178876, 197958, 219076, 242457]
    assert graphs.block_sizes(500000, 5, 2.0).tolist() == [68632, 81617, 97059, 115423, 137269]
    assert graphs.block_sizes(10, 3, 1.0).tolist() == [3, 3, 4]


def test_dsbm_block_counts_within_3_sigma_of_the_reference_generator():
    g = load_golden("sbm_stats")
    k, n = int(g["k"]), int(g["dsbm_n"])
    assert np.allclose(graphs.cyclic_meta_graph(k, float(g["dsbm_eta"]), 0.5), g["dsbm_meta"])
    counts = []
    for seed in range(SEEDS):
        ei, labels = graphs.dsbm(n, k, float(g["dsbm_p"]), g["dsbm_meta"], float(g["dsbm_size_ratio"]), seed)
        assert (ei[0] != ei[1]).all() and np.unique(ei[0] * n + ei[1]).size == ei.shape[1]     # simple digraph
        assert np.bincount(labels, minlength=k).tolist() == graphs.block_sizes(n, k, float(g["dsbm_size_ratio"])).tolist()
        counts.append(graphs.block_counts(ei, labels, k))
    z = _z(np.mean(counts, 0), np.std(counts, 0, ddof=1), SEEDS, g["dsbm_mean"], g["dsbm_std"], int(g["seeds"]))
    assert z.max() <= 3.0, z.round(2)


def test_sdsbm_counts_and_sign_fractions_within_3_sigma_of_the_reference_generator():
    g = load_golden("sbm_stats")
    k, n = int(g["k"]), int(g["sdsbm_n"])
    assert np.allclose(graphs.signed_cyclic_meta_graph(k, float(g["sdsbm_eta"]), 0.5), g["sdsbm_meta"])
    tot, neg = [], []
    for seed in range(SEEDS):
        ei, sign, labels = graphs.sdsbm(n, k, float(g["sdsbm_p"]), g["sdsbm_meta"], float(g["sdsbm_size_ratio"]),
                                        float(g["sdsbm_eta"]), seed)
        assert set(np.unique(sign).tolist()) <= {-1.0, 1.0}
        # SDSBM.py:64-66 flips exactly floor(E * eta) signs
        expected = np.where(g["sdsbm_meta"][labels[ei[0]], labels[ei[1]]] < 0, -1.0, 1.0)
        assert int((sign != expected).sum()) == int(ei.shape[1] * float(g["sdsbm_eta"]))
        tot.append(graphs.block_counts(ei, labels, k))
        neg.append(graphs.block_counts(ei[:, sign < 0], labels, k))
    ref_n = int(g["seeds"])
    z_tot = _z(np.mean(tot, 0), np.std(tot, 0, ddof=1), SEEDS, g["sdsbm_mean"], g["sdsbm_std"], ref_n)
    z_neg = _z(np.mean(neg, 0), np.std(neg, 0, ddof=1), SEEDS, g["sdsbm_neg_mean"], g["sdsbm_neg_std"], ref_n)
    assert z_tot.max() <= 3.0, z_tot.round(2)
    assert z_neg.max() <= 3.0, z_neg.round(2)


def test_ssbm_sign_counts_within_3_sigma_of_the_reference_generator():
    g = load_golden("sbm_stats")
    k, n = int(g["k"]), int(g["ssbm_n"])
    rows = []
    for seed in range(SEEDS):
        ei, sign, labels = graphs.ssbm(n, k, float(g["ssbm_p"]), float(g["ssbm_eta"]), float(g["ssbm_size_ratio"]), seed)
        same = labels[ei[0]] == labels[ei[1]]
        pos = sign > 0
        rows.append([int((pos & same).sum()), int((pos & ~same).sum()), int((~pos & same).sum()), int((~pos & ~same).sum())])
        # both orientations of every pair are stored with the same sign (SSBM.py:102-135)
        fwd = {(int(a), int(b)): float(s) for a, b, s in zip(ei[0][:200], ei[1][:200], sign[:200])}
        lookup = {(int(a), int(b)): float(s) for a, b, s in zip(ei[0], ei[1], sign)}
        assert all(lookup[(b, a)] == s for (a, b), s in fwd.items())
    z = _z(np.mean(rows, 0), np.std(rows, 0, ddof=1), SEEDS, g["ssbm_mean"], g["ssbm_std"], int(g["seeds"]))
    assert z.max() <= 3.0, z.round(2)


def test_edge_targets_of_the_benchmark_configs():
    """p is solved so that E[#edges] hits the BASELINE config sizes (p = 4.0e-4 at 100k / 2M)."""
    ei, _, p = graphs.dsbm_for_edges(100000, 2000000, seed=0)
    assert abs(p - 4.0e-4) < 2e-6 and abs(ei.shape[1] - 2000000) < 5 * 2000000 ** 0.5
    ei, sign, _, p = graphs.sdsbm_for_edges(50000, 1000000, seed=0)
    assert abs(ei.shape[1] - 1000000) < 5 * 1000000 ** 0.5 and 0.3 < float((sign < 0).mean()) < 0.7
