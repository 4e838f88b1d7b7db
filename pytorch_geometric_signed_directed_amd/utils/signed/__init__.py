from .cut_losses import Prob_Balanced_Normalized_Loss, Prob_Balanced_Ratio_Loss, Unhappy_Ratio  # noqa: F401
from .create_spectral_features import create_spectral_features  # noqa: F401
from .link_sign_loss import (Link_Sign_Entropy_Loss, Link_Sign_Product_Loss, Sign_Direction_Loss,  # noqa: F401
                             Sign_Product_Entropy_Loss, Sign_Structure_Loss, Sign_Triangle_Loss, negative_sampling,
                             structured_negative_sampling)
