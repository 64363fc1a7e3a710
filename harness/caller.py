"""The caller's sequence around the hot path, as a standalone harness.

`render_gaussians` restates what `StreetGaussianRenderer.render_kernel_gsplat` does with the six
operators (street_gaussian/models/street_gaussian_renderer.py:186-302): same five operator
calls in the same order with the same keyword arguments, and the same torch glue in between
(opacity * compensation :235-238, dirs/masks :256-258, clamp_min(colors+0.5) :260, depth as 4th
channel :265-266, depth normalisation / clamp / permute :282-300).  It goes through the
`gsplat.rendering` names, i.e. through the drop-in boundary, so benchmarks and parity tests
exercise exactly what the reference would call.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from street_crafter_amd.scenes import Camera, Scene


def render_gaussians(scene: Scene, camera: Camera, tile_size: int = 16, use_depth: bool = True,
                     absgrad: bool = True, antialiasing: bool = True, mode: str = "eval",
                     return_intermediates: bool = False, stage_events=None) -> Dict[str, torch.Tensor]:
    """`stage_events`: optional dict; when given, a (start, end) pair of torch.cuda.Event (HIP events
    on the current stream, the stream every kernel is launched on) is appended per operator call."""
    from gsplat.rendering import (fully_fused_projection, isect_offset_encode, isect_tiles,
                                  rasterize_to_pixels, spherical_harmonics)

    def timed(name, fn, *a, **k):
        if stage_events is None:
            return fn(*a, **k)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn(*a, **k)
        e.record()
        stage_events.setdefault(name, []).append((s, e))
        return r

    xyz3 = scene.means
    rgb3 = scene.sh
    width, height = camera.width, camera.height
    w2c = camera.viewmat.to(xyz3)[None]
    K = camera.K.to(xyz3)[None]

    radii, means2d, depths, conics, compensations = timed(
        "projection", fully_fused_projection, xyz3, None, scene.quats, scene.scales, w2c, K, width, height, packed=False,
        near_plane=camera.znear, far_plane=camera.zfar, calc_compensations=antialiasing)
    opacities = scene.opacities[None, :, 0]
    if compensations is not None:
        opacities = opacities * compensations

    tile_width = math.ceil(width / float(tile_size))
    tile_height = math.ceil(height / float(tile_size))
    tiles_per_gauss, isect_ids, flatten_ids = timed(
        "isect_tiles", isect_tiles, means2d, radii, depths, tile_size, tile_width, tile_height, packed=False, n_cameras=1)
    isect_offsets = timed("isect_offset_encode", isect_offset_encode, isect_ids, 1, tile_width, tile_height)

    dirs = xyz3[None, :, :] - camera.camera_center.to(xyz3)
    masks = radii > 0
    shs = rgb3.expand(1, -1, -1, -1)
    colors = timed("spherical_harmonics", spherical_harmonics, scene.sh_degree, dirs, shs, masks=masks)
    colors = torch.clamp_min(colors + 0.5, 0.0)

    if mode == "train" and means2d.requires_grad:
        means2d.retain_grad()

    if use_depth:
        colors = torch.cat((colors, depths[..., None]), dim=-1)
    if stage_events is not None and hasattr(flatten_ids, "plain"):
        # (probe frames only: the deferred list is settled -- the host's wait for the frame's counts -- BEFORE the rasterizer's
        #  start event, so that the bracket holds the kernel, not the GPU idling while the host waits; the operator would do
        #  the same settle as its first step)
        flatten_ids = flatten_ids.plain()
    render_colors, render_alphas = timed(
        "rasterize_to_pixels", rasterize_to_pixels, means2d, conics, colors, opacities, width, height, tile_size, isect_offsets, flatten_ids,
        backgrounds=None, packed=False, absgrad=absgrad)

    if use_depth:
        rendered_color = render_colors[..., :-1]
        rendered_depth = render_colors[..., -1:] / render_alphas.clamp(min=1e-10)
    else:
        rendered_color = render_colors
        rendered_depth = render_alphas
    rendered_acc = render_alphas
    if mode != "train":
        rendered_color = torch.clamp(rendered_color, 0.0, 1.0)

    result = {
        "rgb": rendered_color[0].permute(2, 0, 1),
        "acc": rendered_acc[..., 0],
        "depth": rendered_depth[..., 0],
        "viewspace_points": means2d,
        "visibility_filter": radii[0] > 0,
        "radii": radii[0] / float(max(height, width)),
    }
    if return_intermediates:
        result.update(dict(_radii=radii, _means2d=means2d, _depths=depths, _conics=conics,
                           _compensations=compensations, _opacities=opacities,
                           _tiles_per_gauss=tiles_per_gauss, _isect_ids=isect_ids,
                           _flatten_ids=flatten_ids, _isect_offsets=isect_offsets, _colors=colors,
                           _render_colors=render_colors, _render_alphas=render_alphas))
    return result


def algorithmic_bytes(n_gauss: int, n_isects: int, width: int, height: int, tile_size: int = 16,
                      sh_bases: int = 4) -> int:
    """compulsory HBM bytes per forward frame at the operator boundaries,
    B_alg = (72 + 25 + 12*K + 52) N + 88 I + 24 P + 4 T   (= 197 N + ... for K = 4)."""
    tiles = math.ceil(width / tile_size) * math.ceil(height / tile_size)
    per_gauss = 72 + (12 + 12 * sh_bases + 1 + 12) + 52
    return per_gauss * n_gauss + 88 * n_isects + 24 * width * height + 4 * tiles


def render_novel_view(fg_scene: Scene, sky_scene, camera: Camera, **kw) -> Dict[str, torch.Tensor]:
    """StreetGaussianRenderer.render_novel_view (street_gaussian_renderer.py:136-163) as the reference
    spells it: the foreground pass over every sub-model but the sky (`render_kernel`, :148), then -- when
    the scene has a sky sub-model (`pc.include_sky`, 5 of the 7 shipped configs) -- a second pass over the
    sky Gaussians alone (`render_sky`, :80-93) and `rgb = rgb + rgb_sky * (1 - acc)` (:152), clamp (:159).
    Both passes go through the drop-in operators exactly like `render_gaussians`; the composite is the
    reference's torch expression.  `sky_scene=None`: single pass (include_sky False)."""
    result = render_gaussians(fg_scene, camera, **kw)
    if sky_scene is not None:
        result_sky = render_gaussians(sky_scene, camera, **kw)
        result["rgb"] = result["rgb"] + result_sky["rgb"] * (1 - result["acc"])
        result["_sky"] = result_sky
    result["rgb"] = torch.clamp(result["rgb"], 0.0, 1.0)
    return result


@torch.no_grad()
def render_novel_view_u8(fg_scene: Scene, sky_scene, camera: Camera, rounding: str = "video", out=None,
                         fused: bool = True) -> torch.Tensor:
    """The same frame as `to_uint8_frame(render_novel_view(...)["rgb"])`, bit for bit, as the sharded
    novel-view loop produces it: each pass is ONE call of gsplat's `rasterization()` (fused forward,
    SURVEY 8f-2) and composite + clamp + uint8 conversion are ONE kernel (sc_frame_composite_u8) instead
    of seven torch elementwise passes over the frame.  -> uint8 [H,W,3]."""
    from gsplat.rendering import rasterization
    from street_crafter_amd.dist import to_uint8_frame

    def one_pass(sc):
        if not fused:
            o = render_gaussians(sc, camera, return_intermediates=True)
            return o["_render_colors"], o["_render_alphas"]
        rc, ra, _ = rasterization(sc.means, sc.quats, sc.scales, sc.opacities.reshape(-1), sc.sh,
                                  camera.viewmat[None], camera.K[None], camera.width, camera.height,
                                  near_plane=camera.znear, far_plane=camera.zfar, sh_degree=sc.sh_degree,
                                  render_mode="RGB+ED", rasterize_mode="antialiased",
                                  camera_centers_=camera.camera_center[None])
        return rc, ra

    rc, ra = one_pass(fg_scene)
    rgb = rc[0, ..., :3].permute(2, 0, 1)
    if sky_scene is None:
        return to_uint8_frame(rgb, rounding=rounding, out=out)
    rc_s, _ = one_pass(sky_scene)
    return to_uint8_frame(rgb, acc=ra[0, ..., 0], sky_rgb_chw=rc_s[0, ..., :3].permute(2, 0, 1),
                          rounding=rounding, out=out)
