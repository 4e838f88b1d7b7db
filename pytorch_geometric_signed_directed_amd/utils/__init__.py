from .directed import Prob_Imbalance_Loss, get_magnetic_Laplacian  # noqa: F401
from .general import get_magnetic_signed_Laplacian  # noqa: F401
from ._norm import add_remaining_self_loops, conv_norm_rw, gcn_norm  # noqa: F401
from .signed import Prob_Balanced_Normalized_Loss, Prob_Balanced_Ratio_Loss, Unhappy_Ratio  # noqa: F401
