"""MI355X-native implementation of StreetCrafter's novel-view rendering hot path.

`street_crafter_amd.rendering` mirrors the `gsplat.rendering` operators and
`street_crafter_amd.knn.distCUDA2` mirrors `simple_knn._C.distCUDA2`; the top-level packages
`gsplat/` and `simple_knn/` in this repository re-export them under the names the reference
imports.  All compute is hand-written HIP for gfx950 behind the C ABI of
include/street_crafter_amd.h.
"""
__version__ = "0.1.0"
