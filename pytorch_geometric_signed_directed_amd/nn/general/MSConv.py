"""MSConv -- drop-in for torch_geometric_signed_directed/nn/general/MSConv.py:12 (signed magnetic
Laplacian, MSGNN).  Same dataflow as MagNetConv on the signed operator."""
from .._magnetic import MagneticChebConv


class MSConv(MagneticChebConv):
    r"""Magnetic signed Laplacian convolution (MSGNN, arXiv 2209.00546).  Args mirror the
    reference (MSConv.py:42-43): ..., normalization='sym', bias=True, cached=False,
    absolute_degree=True."""
    _signed = True

    def __init__(self, in_channels: int, out_channels: int, K: int, q: float, trainable_q: bool,
                 normalization: str = 'sym', bias: bool = True, cached: bool = False,
                 absolute_degree: bool = True, **kwargs):
        operator_memo = kwargs.pop('operator_memo', None)      # memo.py: False = always rebuild like the reference
        kwargs.setdefault('aggr', 'add')
        super().__init__(**kwargs)
        self.absolute_degree = absolute_degree
        self._init_common(in_channels, out_channels, K, q, trainable_q, normalization, cached, bias, operator_memo)
