from .SGCNConv import SGCNConv  # noqa: F401
from .SIMPA import SIMPA  # noqa: F401
from .GATConv import GATConv, SDRLayer  # noqa: F401
from .SNEAConv import SNEAConv  # noqa: F401
