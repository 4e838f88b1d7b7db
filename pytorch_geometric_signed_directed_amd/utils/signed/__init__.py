from .cut_losses import Prob_Balanced_Normalized_Loss, Prob_Balanced_Ratio_Loss, Unhappy_Ratio  # noqa: F401
