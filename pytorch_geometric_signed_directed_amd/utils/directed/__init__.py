from .get_magnetic_Laplacian import get_magnetic_Laplacian  # noqa: F401
from .prob_imbalance_loss import Prob_Imbalance_Loss  # noqa: F401
from .get_adjs_DiGCN import (cal_fast_appr, fast_appr_power, get_appr_directed_adj,  # noqa: F401
                             get_second_directed_adj)
