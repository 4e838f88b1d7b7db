from .MagNetConv import MagNetConv  # noqa: F401
from .DiGCNConv import DiGCNConv  # noqa: F401
from .DGCNConv import DGCNConv  # noqa: F401
from .DIMPA import DIMPA  # noqa: F401
from .complex_relu import complex_relu_layer  # noqa: F401
from .DiGCL import DiGCL, DiGCL_Encoder, GCNConv  # noqa: F401
