from .get_magnetic_signed_Laplacian import get_magnetic_signed_Laplacian  # noqa: F401
