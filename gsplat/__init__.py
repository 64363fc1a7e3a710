"""Drop-in `gsplat` package for StreetCrafter on MI355X.

The reference imports `from gsplat.rendering import rasterization, fully_fused_projection,
isect_tiles, isect_offset_encode, rasterize_to_pixels, spherical_harmonics`
(street_gaussian/models/street_gaussian_renderer.py:204).  Putting this repository on
PYTHONPATH makes that import resolve to the HIP implementation in `street_crafter_amd`.
"""
from .rendering import (fully_fused_projection, isect_offset_encode, isect_tiles,  # noqa: F401
                        rasterization, rasterize_to_pixels, spherical_harmonics)

__version__ = "1.4.0+street_crafter_amd"
