"""Fused rasterization() forward with and without the packed 48-B rasterizer records: frames/s, S-1M and street-1M.
Usage: python tools/exp_packed.py [frames]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gsplat.rendering import rasterization  # noqa: E402
from street_crafter_amd import rendering  # noqa: E402
from street_crafter_amd.scenes import make_scene, make_street_scene  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = "cuda"
W, H = 1920, 1280
cams = [bench.frame_camera(s, W, H).to(dev) for s in range(frames + 10)]


def run(sc):
    op1 = sc.opacities[:, 0].contiguous()
    outs = []
    with torch.no_grad():
        for f, cam in enumerate(cams):
            if f == 10:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            rc, ra, meta = rasterization(sc.means, sc.quats, sc.scales, op1, sc.sh, cam.viewmat[None], cam.K[None], W, H,
                                         near_plane=cam.znear, far_plane=cam.zfar, sh_degree=sc.sh_degree,
                                         render_mode="RGB+ED", rasterize_mode="antialiased")
            if f in (3, 7):
                outs.append((rc.clone(), ra.clone(), meta["conics"].clone(), meta["opacities"].clone(), meta["colors"].clone()))
        torch.cuda.synchronize()
    return frames / (time.perf_counter() - t0), outs


for name, sc in (("S-1M", make_scene(1_000_000).to(dev)), ("street-1M", make_street_scene(1_000_000)[0].to(dev))):
    res = {}
    for rep in range(2):
        for on in (False, True):
            rendering.set_packed_records(on)
            fps, outs = run(sc)
            res.setdefault(on, []).append((fps, outs))
    same = all(all(torch.equal(x, y) for x, y in zip(a, b)) for a, b in zip(res[False][0][1], res[True][0][1]))
    print(f"{name}: four arrays {[round(r[0], 1) for r in res[False]]} frames/s | packed records "
          f"{[round(r[0], 1) for r in res[True]]} frames/s | identical frames and meta: {same}", flush=True)
rendering.set_packed_records(True)
