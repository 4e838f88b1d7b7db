"""TEST INFRASTRUCTURE ONLY -- not shipped, never imported by the product package.

Minimal restatement of the ~20 public `torch_geometric` symbols that
SherylHYX/pytorch_geometric_signed_directed calls on its message-passing path.
`torch_geometric` is an un-vendored, un-pinned dependency of the reference
(reference setup.py:12) and is absent from this image, so the reference cannot
be imported as shipped.  This shim restates PyG's *published* semantics (PyG
>= 2.3 documentation: MessagePassing.propagate = gather -> message -> scatter
-> update; utils.coalesce/scatter/*_self_loops; nn.conv.gcn_conv.gcn_norm) so
that `oracle/gen_golden.py` can execute the reference's own Python, unmodified,
from /root/reference inside this container and record golden vectors.

Every golden written through this shim is additionally cross-checked against an
independent float64 dense-matrix evaluation of the layer formulas
(oracle/dense_f64.py) before it is committed, so a shim mistake cannot
silently define truth.
"""
__version__ = "0.0-shim"
from . import typing, utils, nn, data, datasets  # noqa: F401,E402
