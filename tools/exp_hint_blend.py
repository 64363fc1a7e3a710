"""The rasterizer's work hint under camera motion: neighbourhood weight (sc_set_option raster_hint_blend) and
split threshold, with bench.py's per-frame camera jitter and with a camera that stands still.
Usage: python tools/exp_hint_blend.py [frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from street_crafter_amd import _lib, rendering  # noqa: E402
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_scene, make_street_scene  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = "cuda"
W, H = 1920, 1280
moving = [bench.frame_camera(s, W, H).to(dev) for s in range(frames + 4)]
still = [moving[0]] * (frames + 4)
scenes = {"S-1M": make_scene(1_000_000).to(dev), "street-1M": make_street_scene(1_000_000)[0].to(dev)}


def run(sc, cams):
    ev = {}
    with torch.no_grad():
        for f, cam in enumerate(cams):
            render_gaussians(sc, cam, stage_events=ev if f >= 4 else None)
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev["rasterize_to_pixels"])
    return t[len(t) // 2] * 1e3


for name, sc in scenes.items():
    print("==", name)
    for label, cams in (("moving camera (bench jitter)", moving), ("camera stands still", still)):
        rendering.set_tile_order(False)
        base = run(sc, cams)
        rendering.set_tile_order(True)
        row = [f"plain dispatch {base:6.1f}"]
        for blend in (0, 2, 3, 4):
            _lib.set_option("raster_hint_blend", blend)
            for split in (0, 50):
                _lib.set_option("raster_split", split)
                row.append(f"blend {blend}/4 split {split}: {run(sc, cams):6.1f}")
        _lib.set_option("raster_hint_blend", 3)
        _lib.set_option("raster_split", 50)
        print(f"  {label:30s} raster p50 us: " + " | ".join(row), flush=True)
