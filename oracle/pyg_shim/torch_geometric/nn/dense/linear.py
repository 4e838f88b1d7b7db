import torch


class Linear(torch.nn.Linear):
    """PyG Linear with default initialisers == torch.nn.Linear (kaiming a=sqrt(5))."""
    def __init__(self, in_channels, out_channels, bias=True, **kw):
        super().__init__(in_channels, out_channels, bias=bias)
