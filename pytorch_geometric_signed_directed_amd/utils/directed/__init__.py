from .get_magnetic_Laplacian import get_magnetic_Laplacian  # noqa: F401
