"""pytorch_geometric_signed_directed_amd -- the sparse message-passing hot path of
SherylHYX/pytorch_geometric_signed_directed (MagNetConv / MSConv / DiGCNConv / DGCNConv /
Conv_Base + SIMPA / DIMPA / SGCNConv), built MI355X-native: Python host code over a C-ABI HIP
library (csrc/libpygsd_hip.so, include/pygsd_hip.h).  No torch_geometric dependency, no CPU
fallback: the layers raise if the HIP library is missing or if they are handed CPU tensors."""
__version__ = "0.1.0"

from . import _cabi  # noqa: F401
from .message_passing import MessagePassing  # noqa: F401
from .sparse import Pattern, spmm, spmm2  # noqa: F401
from . import nn, utils  # noqa: F401
