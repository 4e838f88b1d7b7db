import math


def glorot(tensor):
    if tensor is not None:
        a = math.sqrt(6.0 / (tensor.size(-2) + tensor.size(-1)))
        tensor.data.uniform_(-a, a)


def zeros(tensor):
    if tensor is not None:
        tensor.data.fill_(0)
