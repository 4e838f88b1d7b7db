"""Test infrastructure for the checks at the BASELINE configs' STATED sizes: the big synthetic inputs are generated
once per session, kept as .npy files under the system temp directory and memory-mapped by whoever needs them -- the
parent test process, the eight rank processes of a sharded run, the next test -- instead of being re-sampled by each
(the DSBM sampler needs ~15 s of one host core for 20 M edges).  Everything here is data; the float64 references are
computed by oracle/sparse_f64_torch.py."""
import os
import tempfile

import numpy as np
import torch

CACHE = os.path.join(tempfile.gettempdir(), f"pygsd_test_cache_{os.getuid()}")


def _path(name):
    os.makedirs(CACHE, exist_ok=True)
    return os.path.join(CACHE, name + ".npy")


def _cached(name, make):
    """Path of <name>.npy, generated with make() -> ndarray on first use (written atomically)."""
    path = _path(name)
    if not os.path.exists(path):
        arr = make()
        tmp = path + f".{os.getpid()}.tmp.npy"
        np.save(tmp, arr)
        os.replace(tmp, path)
    return path


def load(path):
    return np.load(path, mmap_mode="r")


def dsbm_graph(n, e, seed=0):
    """-> path of the [2, E] int64 edge list (graphs.dsbm_for_edges)."""
    from pytorch_geometric_signed_directed_amd import graphs
    return _cached(f"dsbm_{n}_{e}_{seed}", lambda: graphs.dsbm_for_edges(n, e, seed=seed)[0])


def sdsbm_graph(n, e, seed=1):
    """-> (path of edge list, path of float32 signs) (graphs.sdsbm_for_edges)."""
    from pytorch_geometric_signed_directed_amd import graphs
    pe, ps = _path(f"sdsbm_{n}_{e}_{seed}_ei"), _path(f"sdsbm_{n}_{e}_{seed}_sign")
    if not (os.path.exists(pe) and os.path.exists(ps)):
        ei, sign, _, _ = graphs.sdsbm_for_edges(n, e, seed=seed)
        for path, arr in ((pe, ei), (ps, sign.astype(np.float32))):
            tmp = path + f".{os.getpid()}.tmp.npy"
            np.save(tmp, arr)
            os.replace(tmp, path)
    return pe, ps


def ssbm_graph(n, entries, seed=2):
    """-> (path of the [2, M] edge list holding both orientations, path of the float32 signs): graphs.ssbm with p chosen
    for `entries` stored entries (BASELINE config C3: 500k nodes / 10M +- entries)."""
    from pytorch_geometric_signed_directed_amd import graphs
    pe, ps = _path(f"ssbm_{n}_{entries}_{seed}_ei"), _path(f"ssbm_{n}_{entries}_{seed}_sign")
    if not (os.path.exists(pe) and os.path.exists(ps)):
        p = (entries / 2) / (n * (n - 1) / 2)
        ei, sign, _ = graphs.ssbm(n, 5, p, 0.1, 2.0, seed=seed)
        for path, arr in ((pe, ei), (ps, sign.astype(np.float32))):
            tmp = path + f".{os.getpid()}.tmp.npy"
            np.save(tmp, arr)
            os.replace(tmp, path)
    return pe, ps


def features(n, f, seed, count):
    """-> paths of `count` N(0, 1) float32 [n, f] matrices drawn from one torch generator."""
    names = [f"feat_{n}_{f}_{seed}_{k}" for k in range(count)]
    if not all(os.path.exists(_path(nm)) for nm in names):
        g = torch.Generator().manual_seed(seed)
        for nm in names:
            _cached(nm, lambda: torch.randn(n, f, generator=g).numpy())
    return [_path(nm) for nm in names]


def digcn_operators(n, e, seed=3):
    """Two symmetric, positively weighted, sym-normalised operators with self loops on a DSBM pattern (SURVEY.md 8(d)
    "C5 operators": the reference pre-processing get_adjs_DiGCN.py:113-254 is dense and cannot run at 2M nodes) --
    built on the GPU with plain tensor ops, as tools/bench_configs.py does.  -> [(edge_index path, weight path)] * 2."""
    names = [(f"digcn_{n}_{e}_{seed}_{k}_ei", f"digcn_{n}_{e}_{seed}_{k}_w") for k in range(2)]
    if not all(os.path.exists(_path(a)) and os.path.exists(_path(b)) for a, b in names):
        dev = torch.device("cuda:0")
        src, dst = torch.from_numpy(np.ascontiguousarray(load(dsbm_graph(n, e, seed)))).to(dev)
        loops = torch.arange(n, device=dev)
        for k, (na, nb) in enumerate(names):
            g = torch.Generator(device="cuda").manual_seed(10 + k)
            if k:
                s, d = src[torch.randperm(src.numel(), device=dev, generator=g)], \
                    dst[torch.randperm(dst.numel(), device=dev, generator=g)]
            else:
                s, d = src, dst
            wv = torch.rand(s.numel(), device=dev, generator=g)
            ei = torch.stack([torch.cat([s, d, loops]), torch.cat([d, s, loops])])
            w = torch.cat([wv, wv, torch.ones(n, device=dev)])
            deg = torch.zeros(n, device=dev).index_add_(0, ei[0], w)
            w = deg[ei[0]].rsqrt() * w * deg[ei[1]].rsqrt()
            _cached(na, lambda: ei.cpu().numpy())
            _cached(nb, lambda: w.cpu().numpy())
            del ei, w, wv, s, d
        torch.cuda.empty_cache()
    return [(_path(a), _path(b)) for a, b in names]
