"""A/B of the rasterizer's block -> tile map on S-1M and on the street-shaped scene, interleaved in ONE
process; also checks that both give the bit-identical image.  (Round 2 also tried a heavy-first work list
with half- / quarter-tile work items here: profiles/r02_raster_policy_ab.txt.)
Usage: python tools/exp_raster.py [frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from street_crafter_amd import _lib  # noqa: E402
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene  # noqa: E402

DEFAULT = dict(raster_map=1)
POLICIES = [
    ("one band of tile rows per XCD (round 1)", dict(raster_map=0)),
    ("neighbouring tiles round-robin over the XCDs", dict(raster_map=1)),
]


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = "cuda"
    cam = make_camera().to(dev)
    fg, sky = make_street_scene(1_000_000)
    scenes = {"S-1M": make_scene(1_000_000).to(dev), "street-1M": fg.to(dev), "street sky 31k": sky.to(dev)}
    for name, sc in scenes.items():
        ref = None
        print(f"== {name}")
        for label, opts in POLICIES:
            for k, v in {**DEFAULT, **opts}.items():
                _lib.set_option(k, v)
            ev = {}
            with torch.no_grad():
                for f in range(frames + 2):
                    out = render_gaussians(sc, cam, stage_events=ev if f >= 2 else None, return_intermediates=True)
            torch.cuda.synchronize()
            t = sorted(a.elapsed_time(b) for a, b in ev["rasterize_to_pixels"])
            img = out["_render_colors"].clone()
            same = True if ref is None else bool(torch.equal(ref.view(torch.int32), img.view(torch.int32)))
            ref = img if ref is None else ref
            print(f"  {label:55s} raster p50 {t[len(t) // 2] * 1e3:8.1f} us  min {t[0] * 1e3:8.1f} us   identical={same}"
                  f"   I={out['_isect_ids'].numel()}", flush=True)
        del ref
    for k, v in DEFAULT.items():
        _lib.set_option(k, v)


if __name__ == "__main__":
    main()
