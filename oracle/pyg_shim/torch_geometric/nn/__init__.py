from . import inits, conv, dense  # noqa: F401
from .conv import MessagePassing, GATConv, GCNConv  # noqa: F401
