from .linear import Linear  # noqa: F401
