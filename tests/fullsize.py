"""Test infrastructure of tests/test_gpu_fullsize.py: inputs, parameters and float64 references (evaluated on the device by
oracle/sparse_f64_torch.py) of the BASELINE configurations at their STATED sizes."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import bigdata  # noqa: E402

WORLD = 8
N_SAMPLE = 1024
MAGNETIC = {
    "northstar": dict(n=1000000, e=20000000, h=64, k=1, signed=False),
    "c4": dict(n=1000000, e=20000000, h=128, k=2, signed=True),
}
RUNS = [("northstar", "auto"), ("northstar", "rows"), ("c4", "auto"), ("c4", "rows")]
C5 = dict(n=2000000, e=25000000, h=64)
BF16_TOL = 3 * 2.0 ** -8


def sample_rows(n):
    return np.sort(np.random.default_rng(3).choice(n, N_SAMPLE, replace=False))


def magnetic_files(name):
    cfg = MAGNETIC[name]
    if cfg["signed"]:
        p_ei, p_sign = bigdata.sdsbm_graph(cfg["n"], cfg["e"], seed=1)
    else:
        p_ei, p_sign = bigdata.dsbm_graph(cfg["n"], cfg["e"], seed=0), None
    return p_ei, p_sign, bigdata.features(cfg["n"], cfg["h"], 3, 4)


def magnetic_params(name):
    """Layer parameters of a configuration, identical wherever they are built (seeded)."""
    from pytorch_geometric_signed_directed_amd.nn import MagNetConv, MSConv
    cfg = MAGNETIC[name]
    torch.manual_seed(11)
    proto = (MSConv if cfg["signed"] else MagNetConv)(cfg["h"], cfg["h"], cfg["k"], 0.25, False)
    with torch.no_grad():
        proto.bias.uniform_(-0.5, 0.5)
    return proto.weight.detach().clone(), proto.bias.detach().clone()


def inception_params():
    from pytorch_geometric_signed_directed_amd.nn import DiGCN_InceptionBlock
    torch.manual_seed(6)
    return DiGCN_InceptionBlock(C5["h"], C5["h"]).state_dict()


def dev_tensor(path, dev, dtype=None):
    t = torch.from_numpy(np.array(bigdata.load(path))).to(dev)
    return t if dtype is None else t.to(dtype)


def errors(got, want):
    d = (got.double() - want.double()).abs()
    top = float(want.abs().max())
    return {"max_abs_err": float(d.max()), "max_mixed_err": float((d / (1.0 + want.double().abs())).max()),
            "max_norm_rel_err": float(d.max()) / max(1.0, top), "max_abs_want": top}


def magnetic_reference(name, dev):
    """float64 (out_real, out_imag, dx_real, dx_imag, dW, db) of <out_real, g_real> + <out_imag, g_imag>."""
    from oracle import sparse_f64_torch as T64
    cfg = MAGNETIC[name]
    p_ei, p_sign, feats = magnetic_files(name)
    weight, bias = magnetic_params(name)
    ei = dev_tensor(p_ei, dev)
    sign = None if p_sign is None else dev_tensor(p_sign, dev)
    op = T64.magnetic_operator(ei, sign, cfg["n"], 0.25, signed=cfg["signed"], absolute_degree=True)
    xr, xi, gr, gi = (dev_tensor(p, dev, torch.float64) for p in feats)
    return T64.magnet_conv(xr, xi, op, weight.to(dev), bias.to(dev), gr, gi)


def inception_reference(dev):
    """float64 on bf16-ROUNDED inputs and parameters, loss = sum_k (k + 1) <x_k, g>: (outs [3], dx, param grads)."""
    from oracle import sparse_f64_torch as T64
    n, h = C5["n"], C5["h"]
    ops = bigdata.digcn_operators(n, C5["e"], seed=3)
    p_x, p_g = bigdata.features(n, h, 6, 2)
    rnd = (lambda t: t.to(torch.bfloat16).double())
    sd = {k: rnd(v.detach().to(dev)) for k, v in inception_params().items()}
    x, go = rnd(dev_tensor(p_x, dev)), dev_tensor(p_g, dev, torch.float64)
    outs, dx, grads = [x @ sd["ln.weight"].t() + sd["ln.bias"]], go @ sd["ln.weight"], {}
    grads["ln.weight"], grads["ln.bias"] = go.t() @ x, go.sum(0)
    for k, name in ((1, "conv1"), (2, "conv2")):
        ei, w = dev_tensor(ops[k - 1][0], dev), dev_tensor(ops[k - 1][1], dev)
        o, dxk, dwk, dbk = T64.digcn_conv(x, ei, w, sd[name + ".weight"], sd[name + ".bias"], (k + 1.0) * go)
        outs.append(o)
        dx = dx + dxk
        grads[name + ".weight"], grads[name + ".bias"] = dwk, dbk
        del ei, w
    return outs, dx, grads


def shard(path, plan, width, dev):
    """This rank's rows of a cached [n, width] matrix, zero-padded to n_pad rows (only those rows leave the memory map)."""
    t = torch.zeros((plan.n_pad, width), dtype=torch.float32)
    t[:plan.n_local] = torch.from_numpy(np.array(bigdata.load(path)[plan.lo:plan.hi]))
    return t.to(dev)
