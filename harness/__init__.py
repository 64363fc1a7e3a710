"""Measurement / test harness around the product (NOT part of the shipped package).

`harness.caller` restates the reference's CALLER of the hot path (`render_kernel_gsplat` /
`render_novel_view`, street_gaussian/models/street_gaussian_renderer.py:136-163,186-302) on top of the drop-in
`gsplat.rendering` names, so that bench.py, the parity tests and the tools drive the operators exactly as the
reference would.  What ships is `street_crafter_amd/` (operators + C ABI), `gsplat/` and `simple_knn/`.
"""
