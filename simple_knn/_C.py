"""`simple_knn._C.distCUDA2` -> street_crafter_amd HIP kernel (see simple_knn/__init__.py)."""
from street_crafter_amd.knn import distCUDA2  # noqa: F401

__all__ = ["distCUDA2"]
