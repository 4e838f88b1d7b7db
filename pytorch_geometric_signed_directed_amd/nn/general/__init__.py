from .MSConv import MSConv  # noqa: F401
from .conv_base import Conv_Base, conv_norm_rw  # noqa: F401
