"""simple_knn._C.distCUDA2 mirror for MI355X.

`distCUDA2(points f32[N,3] on the GPU) -> f32[N]`: mean squared distance to the 3 nearest other
points, as used at street_gaussian/models/gaussian_model.py:65 (callers clamp to >= 1e-7 and take
log(sqrt())).  Launches the HIP kernels of csrc/knn.hip through the C ABI; no CPU path.
"""
from __future__ import annotations

import torch

from . import _lib


@torch.no_grad()
def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if not isinstance(points, torch.Tensor):
        raise TypeError("points must be a torch.Tensor")
    if not points.is_cuda:
        raise RuntimeError(f"points must live on a HIP device (got {points.device}); "
                           "street_crafter_amd has no CPU path")
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError(f"points must be [N,3], got {tuple(points.shape)}")
    lib = _lib.load()
    pts = points.detach().to(torch.float32).contiguous()
    n = pts.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=pts.device)
    if n == 0:
        return out
    ws = torch.empty(lib.sc_knn_workspace_bytes(n), dtype=torch.uint8, device=pts.device)
    _lib.check(lib.sc_knn3_mean_dist2(pts.data_ptr(), n, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                      torch.cuda.current_stream(pts.device).cuda_stream),
               "sc_knn3_mean_dist2")
    return out
