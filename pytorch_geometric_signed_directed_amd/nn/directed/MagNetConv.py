"""MagNetConv -- drop-in for torch_geometric_signed_directed/nn/directed/MagNetConv.py:13
(same constructor, forward, state_dict keys and __repr__), computed by the fused dual-value HIP
SpMM (see nn/_magnetic.py)."""
from .._magnetic import MagneticChebConv


class MagNetConv(MagneticChebConv):
    r"""The magnetic graph convolutional operator of MagNet (arXiv 2102.11391):
    Chebyshev filter of the scaled, normalised magnetic Laplacian 2L/lambda_max - I.

    Args mirror the reference (MagNetConv.py:44-45): in_channels, out_channels, K, q, trainable_q,
    normalization ('sym' | None), cached, bias.
    """

    def __init__(self, in_channels: int, out_channels: int, K: int, q: float, trainable_q: bool,
                 normalization: str = 'sym', cached: bool = False, bias: bool = True, **kwargs):
        operator_memo = kwargs.pop('operator_memo', None)      # memo.py: False = always rebuild like the reference
        kwargs.setdefault('aggr', 'add')
        super().__init__(**kwargs)
        # the reference sets flow='target_to_source' AFTER super().__init__ (MagNetConv.py:51),
        # which is a no-op: the effective flow is source_to_target
        self._init_common(in_channels, out_channels, K, q, trainable_q, normalization, cached, bias, operator_memo)
