from .get_magnetic_Laplacian import get_magnetic_Laplacian  # noqa: F401
from .prob_imbalance_loss import Prob_Imbalance_Loss  # noqa: F401
