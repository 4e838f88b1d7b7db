from .directed import get_magnetic_Laplacian  # noqa: F401
from .general import get_magnetic_signed_Laplacian  # noqa: F401
from ._norm import add_remaining_self_loops, conv_norm_rw, gcn_norm  # noqa: F401
