"""Would running isect_tiles on a side stream, beside the SH kernel and the caller's glue of the SAME frame, pay?
The caller's sequence restated by hand with two streams per frame (an experiment: the product keeps one stream).
Usage: python tools/exp_overlap.py [frames]"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gsplat.rendering import (fully_fused_projection, isect_offset_encode, isect_tiles, rasterize_to_pixels,  # noqa: E402
                              spherical_harmonics)
from street_crafter_amd.dist import to_uint8_frame  # noqa: E402
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_scene  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 80
dev = "cuda"
W, H = 1920, 1280
cams = [bench.frame_camera(s, W, H).to(dev) for s in range(frames + 10)]
sc = make_scene(1_000_000).to(dev)
side = torch.cuda.Stream()
out = torch.empty(H, W, 3, dtype=torch.uint8, device=dev)


def frame_seq(cam):
    with torch.no_grad():
        return to_uint8_frame(render_gaussians(sc, cam)["rgb"], out=out)


def frame_overlap(cam):
    with torch.no_grad():
        main = torch.cuda.current_stream()
        w2c, K = cam.viewmat[None], cam.K[None]
        radii, means2d, depths, conics, comp = fully_fused_projection(
            sc.means, None, sc.quats, sc.scales, w2c, K, W, H, packed=False, near_plane=cam.znear, far_plane=cam.zfar,
            calc_compensations=True)
        side.wait_stream(main)
        tw, th = math.ceil(W / 16), math.ceil(H / 16)
        with torch.cuda.stream(side):
            tpg, ids, fids = isect_tiles(means2d, radii, depths, 16, tw, th, packed=False, n_cameras=1)
            offs = isect_offset_encode(ids, 1, tw, th)
        opac = sc.opacities[None, :, 0] * comp
        dirs = sc.means[None] - cam.camera_center
        masks = radii > 0
        colors = spherical_harmonics(sc.sh_degree, dirs, sc.sh.expand(1, -1, -1, -1), masks=masks)
        colors = torch.clamp_min(colors + 0.5, 0.0)
        colors = torch.cat((colors, depths[..., None]), dim=-1)
        main.wait_stream(side)
        rc, ra = rasterize_to_pixels(means2d, conics, colors, opac, W, H, 16, offs, fids, backgrounds=None, packed=False,
                                     absgrad=True)
        rgb = torch.clamp(rc[..., :-1], 0.0, 1.0)
        _ = rc[..., -1:] / ra.clamp(min=1e-10)
        for t in (means2d, radii, depths, fids, offs):
            t.record_stream(side); t.record_stream(main)
        return to_uint8_frame(rgb[0].permute(2, 0, 1), out=out)


ref = frame_seq(cams[0]).clone()
got = frame_overlap(cams[0]).clone()
torch.cuda.synchronize()
print("identical frame:", bool(torch.equal(ref, got)))
for name, fn in (("caller sequence, one stream", frame_seq), ("isect_tiles on a side stream", frame_overlap),
                 ("caller sequence, one stream", frame_seq), ("isect_tiles on a side stream", frame_overlap)):
    for s in range(10):
        fn(cams[s])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(10, 10 + frames):
        fn(cams[s])
    torch.cuda.synchronize()
    print(f"  {name:32s} {(time.perf_counter() - t0) / frames * 1e3:.3f} ms/frame", flush=True)
