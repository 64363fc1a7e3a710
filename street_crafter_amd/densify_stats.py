"""Densification statistics: the consumer of the rasterizer's backward side outputs.

Host-side torch mirror of what the reference's training loop does with `viewspace_points.grad`,
`viewspace_points.absgrad`, `visibility_filter` and `radii` after every `loss.backward()`:

    StreetGaussianModel.set_max_radii2D          street_gaussian/models/street_gaussian_model.py:487-497
    StreetGaussianModel.add_densification_stats  street_gaussian/models/street_gaussian_model.py:504-521
    GaussianModelBkgd.densify_and_prune (grads)  street_gaussian/models/gaussian_model_bkgd.py:100-106

It is plain torch (the reference's own code there is plain torch too); it exists so that the
train-path contract of the HIP operators -- `.grad` on a non-leaf projection output, `.absgrad`
attached to the same tensor object, integer `radii` -- is exercised end to end by tests and by the
training benchmark, with the same per-model slicing (`graph_gaussian_range`) the scene graph uses.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch


class DensificationStats:
    """Per sub-model accumulators with the reference's shapes: `xyz_gradient_accum` [n,2]
    (column 0: absgrad norm, column 1: grad norm -- street_gaussian_model.py:518-519), `denom` [n,1],
    `max_radii2D` [n]."""

    def __init__(self, graph_gaussian_range: Dict[str, Tuple[int, int]], device="cuda"):
        self.graph_gaussian_range = dict(graph_gaussian_range)
        self.device = torch.device(device)
        self.xyz_gradient_accum: Dict[str, torch.Tensor] = {}
        self.denom: Dict[str, torch.Tensor] = {}
        self.max_radii2D: Dict[str, torch.Tensor] = {}
        self.reset()

    def reset(self):
        """gaussian_model_bkgd.py:151-153 (after every densify_and_prune)."""
        for name, (start, end) in self.graph_gaussian_range.items():
            n = end - start
            self.xyz_gradient_accum[name] = torch.zeros((n, 2), device=self.device)
            self.denom[name] = torch.zeros((n, 1), device=self.device)
            self.max_radii2D[name] = torch.zeros((n,), device=self.device)

    @torch.no_grad()
    def set_max_radii2D(self, radii: torch.Tensor, visibility_filter: torch.Tensor):
        """radii: what the renderer returns under "radii" (= radii[0] / max(H, W), renderer.py:299)."""
        radii = radii.float()
        for name, (start, end) in self.graph_gaussian_range.items():
            vis = visibility_filter[start:end]
            r = radii[start:end]
            m = self.max_radii2D[name]
            m[vis] = torch.max(m[vis], r[vis])

    @torch.no_grad()
    def add_densification_stats(self, viewspace_point_tensor: torch.Tensor, visibility_filter: torch.Tensor,
                                image_width: int, image_height: int):
        if hasattr(viewspace_point_tensor, "absgrad"):
            g = torch.cat([viewspace_point_tensor.absgrad, viewspace_point_tensor.grad], dim=-1)
            if g.ndim == 3:
                g = g[0]
            g = g * 0.5 * torch.as_tensor([image_width, image_height, image_width, image_height]).to(g)
        else:
            g = viewspace_point_tensor.grad
            if g.ndim == 3:
                g = g[0]
        for name, (start, end) in self.graph_gaussian_range.items():
            vis = visibility_filter[start:end]
            gm = g[start:end]
            acc = self.xyz_gradient_accum[name]
            acc[vis, 0:1] += torch.norm(gm[vis, :2], dim=-1, keepdim=True)
            acc[vis, 1:2] += torch.norm(gm[vis, 2:], dim=-1, keepdim=True)
            self.denom[name][vis] += 1

    @torch.no_grad()
    def mean_grads(self, name: str, use_abs: bool = False) -> torch.Tensor:
        """The quantity compared with `densify_grad_threshold` (gaussian_model_bkgd.py:100-105)."""
        col = 1 if use_abs else 0
        grads = self.xyz_gradient_accum[name][:, col:col + 1] / self.denom[name]
        grads[grads.isnan()] = 0.0
        return grads

    @torch.no_grad()
    def clone_split_masks(self, name: str, max_grad: float, scaling_max: torch.Tensor, extent: float,
                          percent_dense: float = 0.01, use_abs: bool = False
                          ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Which Gaussians the next densification step would clone / split
        (selection rules of gaussian_model.py:493-498 densify_and_clone and :452-463
        densify_and_split: gradient over the threshold, size below / above percent_dense * extent)."""
        g = self.mean_grads(name, use_abs).squeeze(-1)
        hot = g >= max_grad
        small = scaling_max <= percent_dense * extent
        return hot & small, hot & ~small


def accumulate_from_render(stats: DensificationStats, out: Dict[str, torch.Tensor], image_width: int,
                           image_height: int):
    """The two calls train.py makes after backward (train.py:283-284): max radii, then the gradient
    statistics, on the renderer's result dict."""
    stats.set_max_radii2D(out["radii"], out["visibility_filter"])
    stats.add_densification_stats(out["viewspace_points"], out["visibility_filter"], image_width, image_height)
