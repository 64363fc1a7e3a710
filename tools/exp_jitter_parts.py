"""Which part of bench.py's camera jitter costs the rasterizer's work hint: the yaw (image-space shift, the same for
every depth: recoverable by looking the hint up at the shifted tile) or the translation (depth-dependent parallax)?
Usage: python tools/exp_jitter_parts.py [frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = "cuda"
W, H = 1920, 1280
n = frames + 8


def cams(yaw_amp, shift_amp):
    out = []
    for f in range(n):
        j = ((f * 37) % 11 - 5) / 5.0
        out.append(make_camera(W, H, 2050.0, 2050.0, yaw=yaw_amp * j, shift=(shift_amp * j, 0.0, 0.0)).to(dev))
    return out


patterns = {"still": cams(0.0, 0.0), "yaw +-0.01 only": cams(0.01, 0.0), "shift +-0.05 m only": cams(0.0, 0.05),
            "both (bench.py)": cams(0.01, 0.05), "yaw +-0.03 only": cams(0.03, 0.0)}


def run(sc, cs):
    ev = {}
    with torch.no_grad():
        for f, c in enumerate(cs):
            render_gaussians(sc, c, stage_events=ev if f >= 8 else None)
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev["rasterize_to_pixels"])
    return t[len(t) // 2] * 1e3


for name, sc in (("S-1M", make_scene(1_000_000).to(dev)), ("street-1M", make_street_scene(1_000_000)[0].to(dev))):
    run(sc, patterns["still"])
    print(name + ": " + " | ".join(f"{k} {run(sc, v):6.1f}" for k, v in patterns.items()), flush=True)
