"""The rasterizer's work hint when consecutive calls come from DIFFERENT cameras (a street rig: front / front-left /
front-right rendered in turn for every time step): the hint a frame finds is the previous call's, i.e. another view's.
Rasterizer p50 per pattern: plain dispatch, the dispatch list with ONE hint buffer for all views (round 2's first form),
and with the hint kept per view slot.
Usage: python tools/exp_camera_rig.py [frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from street_crafter_amd import rendering  # noqa: E402
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda"
W, H = 1920, 1280
F = 2050.0


def cam(yaw, shift=0.0):
    return make_camera(W, H, F, F, yaw=yaw, shift=(shift, 0.0, 0.0)).to(dev)


n = frames + 6
patterns = {
    "one camera, bench jitter": [bench.frame_camera(s, W, H).to(dev) for s in range(n)],
    "one camera, still": [cam(0.0)] * n,
    "two cameras in turn (yaw +-0.25)": [cam(0.25 if s % 2 else -0.25) for s in range(n)],
    "three cameras in turn (yaw 0, +0.4, -0.4)": [cam((0.0, 0.4, -0.4)[s % 3]) for s in range(n)],
}
scenes = {"S-1M": make_scene(1_000_000).to(dev), "street-1M": make_street_scene(1_000_000)[0].to(dev)}


def run(sc, cams):
    ev = {}
    with torch.no_grad():
        for f, c in enumerate(cams):
            render_gaussians(sc, c, stage_events=ev if f >= 6 else None)
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev["rasterize_to_pixels"])
    return t[len(t) // 2] * 1e3, t[-1] * 1e3


def frames_per_s(sc, cams, hist_len):
    """whole frames (one stream), and how often the speculative sort launch had been sized too small"""
    import time
    rendering._STATE.history.clear()
    rendering._STATE.history_len = hist_len
    with torch.no_grad():
        for c in cams[:6]:
            render_gaussians(sc, c)
        torch.cuda.synchronize()
        s0 = dict(rendering._STATE.stats)
        t0 = time.perf_counter()
        for c in cams[6:]:
            render_gaussians(sc, c)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    rendering._STATE.history_len = 8
    rendering._STATE.history.clear()
    return (len(cams) - 6) / dt, rendering._STATE.stats["exact_relaunch"] - s0["exact_relaunch"]


for name, sc in scenes.items():
    print("==", name)
    for label, cams in patterns.items():
        rendering.set_tile_order(False)
        off, _ = run(sc, cams)
        rendering.set_tile_order(True)
        rendering.set_view_slots(False)
        shared, _ = run(sc, cams)
        rendering.set_view_slots(True)
        on, worst = run(sc, cams)
        print(f"  {label:46s} raster p50 us: plain dispatch {off:6.1f} | list, one hint for all views {shared:6.1f} | "
              f"list, hint per view slot {on:6.1f} (max {worst:6.1f})", flush=True)
        f1, r1 = frames_per_s(sc, cams, 1)
        f8, r8 = frames_per_s(sc, cams, 8)
        print(f"  {'':46s} frames/s, sort sized by the previous call: {f1:7.1f} ({r1} of {len(cams) - 6} relaunched) | by the "
              f"largest of the last 8 calls: {f8:7.1f} ({r8} relaunched)", flush=True)
