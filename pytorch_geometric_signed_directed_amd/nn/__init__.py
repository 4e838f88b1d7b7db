from .directed import *  # noqa: F401,F403
from .general import *  # noqa: F401,F403
from .signed import *  # noqa: F401,F403
from .models import (DGCN_node_classification, DIGRAC_node_clustering, DiGCN_Inception_Block_node_classification,  # noqa: F401
                     DiGCN_InceptionBlock, DiGCN_node_classification, MagNet_link_prediction,
                     MagNet_node_classification, MSGNN_link_prediction, MSGNN_node_classification,
                     SSSNET_node_clustering)
from .models import (DGCN_link_prediction, DiGCN_Inception_Block_link_prediction, DiGCN_link_prediction,  # noqa: F401
                     SDGNN, SGCN, SiGAT, SNEA, SSSNET_link_prediction)
