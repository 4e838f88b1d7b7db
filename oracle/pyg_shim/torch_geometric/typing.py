"""Type aliases the reference imports from torch_geometric.typing."""
from typing import Optional, Tuple, Union  # noqa: F401 (re-exported: SGCNConv.py:6)
from torch import Tensor


class SparseTensor:  # torch_sparse is absent; SparseTensor branches are out of scope
    def __init__(self, *a, **k):
        raise NotImplementedError("torch_sparse.SparseTensor is not available in this image")


Adj = Union[Tensor, SparseTensor]
OptTensor = Optional[Tensor]
PairTensor = Tuple[Tensor, Tensor]
OptPairTensor = Tuple[Tensor, Optional[Tensor]]
