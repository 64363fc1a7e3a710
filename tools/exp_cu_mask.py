"""Frames in flight with the rasterizer on a CU-masked / low-priority side stream.
S-1M through the reference caller sequence -> uint8 frame, 3 frames in flight; every configuration renders the same frames
(compared bit for bit with the plain configuration) and is timed twice (better run reported).

    python tools/exp_cu_mask.py [frames] [configs,comma,separated] [n_gauss]
configs: base | side (unmasked side streams: the price of the hand-over) | low (side streams of the lowest priority) |
         lowhigh (side lowest, frame streams highest) | cuNNN (side streams confined to NNN CUs) |
         cuNNNh (the same + frame streams of the highest priority) | one:<cfg> (ONE side stream shared by all frame streams)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd import rendering  # noqa: E402
from street_crafter_amd.dist import destroy_stream, make_stream, to_uint8_frame  # noqa: E402
from street_crafter_amd.scenes import make_scene  # noqa: E402

FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 60
CONFIGS = (sys.argv[2] if len(sys.argv) > 2 else "base,side,low,lowhigh,cu224,cu192,cu160,cu224h,one:cu224").split(",")
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
NSTR = 3
W, H = 1920, 1280
dev = torch.device("cuda", 0)
scene = make_scene(N).to(dev)
WARM = 9
cams = [bench.frame_camera(s, W, H).to(dev) for s in range(FRAMES + WARM)]
out = torch.empty((FRAMES + WARM, H, W, 3), dtype=torch.uint8, device=dev)


def run(cfg):
    shared = cfg.startswith("one:")
    c = cfg[4:] if shared else cfg
    high = c in ("lowhigh",) or c.endswith("h") and c.startswith("cu")
    mains = [make_stream(dev, priority=-100 if high else None) for _ in range(NSTR)]
    sides = []
    if c != "base":
        for _ in range(1 if shared else NSTR):
            if c.startswith("cu"):
                sides.append(make_stream(dev, cus=int(c[2:].rstrip("h"))))
            elif c in ("low", "lowhigh"):
                sides.append(make_stream(dev, priority=100))
            else:
                sides.append(make_stream(dev))
        for k, m in enumerate(mains):
            rendering.set_raster_side_stream(dev, sides[0 if shared else k], main=m)
    best = None
    try:
        for rep in range(2):
            home = torch.cuda.current_stream(dev)
            torch.cuda.synchronize(dev)
            t0 = None
            with torch.no_grad():
                for s in range(FRAMES + WARM):
                    if s == WARM:
                        torch.cuda.synchronize(dev)
                        t0 = time.perf_counter()
                    torch.cuda.set_stream(mains[s % NSTR])
                    o = render_gaussians(scene, cams[s])
                    to_uint8_frame(o["rgb"], out=out[s])
            torch.cuda.set_stream(home)
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
    finally:
        for m in mains:
            rendering.set_raster_side_stream(dev, None, main=m)
        torch.cuda.synchronize(dev)
        for st in mains + sides:
            destroy_stream(st)
    return best, out[WARM:].clone()


ref = None
for cfg in CONFIGS:
    el, frames = run(cfg)
    same = ""
    if ref is None:
        ref = frames
    else:
        same = "  frames identical to the first configuration: " + str(bool(torch.equal(ref, frames)))
    print(f"{cfg:12s} {el / FRAMES * 1e3:7.4f} ms/frame  {FRAMES / el:7.1f} frames/s{same}", flush=True)
