"""How much of the rasterizer's time is a draining tail?  Times rasterize_to_pixels on S-1M with a fraction of
the tiles masked off (masked tiles return at once): time(n_tiles) = tail + per_tile * n_tiles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsplat.rendering as R  # noqa: E402
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "s1m"
sc = (make_scene(1_000_000) if which == "s1m" else make_street_scene(1_000_000)[0]).to("cuda")
cam = make_camera().to("cuda")
with torch.no_grad():
    o = render_gaussians(sc, cam, return_intermediates=True)
    args = (o["_means2d"], o["_conics"], o["_colors"], o["_opacities"], 1920, 1280, 16, o["_isect_offsets"], o["_flatten_ids"])
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = torch.rand(1, 80, 120, device="cuda", generator=g)
    for keep in (1.0, 0.75, 0.5, 0.25, 0.125, 0.0625):
        masks = rnd < keep
        ts = []
        for _ in range(12):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            R.rasterize_to_pixels(*args, masks=masks)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        print(f"{which}: {int(masks.sum()):5d} tiles  {ts[len(ts) // 2] * 1e3:7.1f} us", flush=True)
