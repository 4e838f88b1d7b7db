from .directed import Prob_Imbalance_Loss, get_magnetic_Laplacian  # noqa: F401
from .general import get_magnetic_signed_Laplacian  # noqa: F401
from ._norm import add_remaining_self_loops, conv_norm_rw, gcn_norm  # noqa: F401
from .signed import (Link_Sign_Entropy_Loss, Link_Sign_Product_Loss, Prob_Balanced_Normalized_Loss,  # noqa: F401
                     Prob_Balanced_Ratio_Loss, Sign_Direction_Loss, Sign_Product_Entropy_Loss, Sign_Structure_Loss,
                     Sign_Triangle_Loss,
                     Unhappy_Ratio, create_spectral_features)
