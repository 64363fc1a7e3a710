"""`gsplat.rendering` names -> street_crafter_amd HIP operators (see gsplat/__init__.py)."""
from street_crafter_amd.rendering import (fully_fused_projection, isect_offset_encode,  # noqa: F401
                                          isect_tiles, rasterization, rasterize_to_pixels,
                                          spherical_harmonics)

__all__ = ["rasterization", "fully_fused_projection", "isect_tiles", "isect_offset_encode",
           "rasterize_to_pixels", "spherical_harmonics"]
