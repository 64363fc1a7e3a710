"""Replays one fuzz case's backward comparison with more detail: the rasterizer's own gradient outputs (means2d,
absgrad) of the wave kernel vs the reference-shaped kernel, with the dispatch list on / off and half tiles on / off.
Usage: python tools/debug_bwd_fuzz.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from street_crafter_amd import _lib, rendering  # noqa: E402
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene  # noqa: E402

W, H, n = 3756, 20, 20000
sc = make_scene(n, sh_degree=1, seed=12345, z_range=(10.0, 400.0), scale_range=(0.004, 3.0)).to("cuda")
f = 2050.0 * W / 1920.0
cam = make_camera(W, H, f, f).to("cuda")
target = torch.rand(3, H, W, device="cuda")


def run(variant, order_on, bwd_split):
    params = [t.detach().clone().requires_grad_(True) for t in (sc.means, sc.quats, sc.scales, sc.opacities, sc.sh)]
    scn = type(sc)(*params, sc.sh_degree)
    p1 = _lib.set_option("raster_bwd", variant)
    p2 = _lib.set_option("raster_bwd_split", bwd_split)
    p3 = rendering.set_tile_order(order_on)
    try:
        for _ in range(2):                      # second pass: warm dispatch list
            for p in params:
                p.grad = None
            out = render_gaussians(scn, cam, mode="train")
            ((out["rgb"] - target).abs().mean() + 0.05 * out["acc"].mean() + 0.01 * out["depth"].mean()).backward()
    finally:
        _lib.set_option("raster_bwd", p1); _lib.set_option("raster_bwd_split", p2); rendering.set_tile_order(p3)
    vp = out["viewspace_points"]
    return dict(means=params[0].grad, quats=params[1].grad, scales=params[2].grad, opac=params[3].grad, sh=params[4].grad,
                means2d=vp.grad, absgrad=vp.absgrad)


ref = run(0, False, 0)
ref2 = run(0, False, 0)
for label, cfg in (("reference-shaped again (run-to-run noise of its own atomics)", None), ("wave kernel, plain dispatch", (1, False, 0)),
                   ("wave kernel, list, whole tiles", (1, True, 0)), ("wave kernel, list, half tiles", (1, True, 1))):
    g = ref2 if cfg is None else run(*cfg)
    row = []
    for k in ref:
        den = float(ref[k].abs().max()) + 1e-30
        row.append(f"{k} {float((g[k] - ref[k]).abs().max()) / den:.2e}")
    print(f"{label:62s}", "  ".join(row), flush=True)
