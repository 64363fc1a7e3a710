"""fwd+bwd step timing of the caller's sequence (BASELINE config 3 shape: train-mode render +
loss.backward()) on a synthetic scene.  Prints ms per step and the backward share.

    python tools/exp_train.py [n_gauss] [width] [height] [steps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1600     # the reference trains at 1600x1066 (camera_utils.py:150-152)
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1066
STEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = "cuda"
if os.environ.get("SC_NO_CAP_CLAMP"):      # A/B: predicted bucket capacity with its head-room, whatever it crosses (before the clamp)
    from street_crafter_amd import rendering as _r
    _r._STATE.bucket_cap["v"] = 1 << 40
if os.environ.get("SC_NATIVE_AUTOGRAD") == "0":      # A/B: the Python torch.autograd.Functions instead of the C++ ones
    from street_crafter_amd import rendering as _r
    _r.set_native_autograd(False)
if os.environ.get("SC_DEFER") == "0":      # A/B: isect_tiles waits for the counts itself
    from street_crafter_amd import rendering as _r
    _r.set_deferred_isect(False)
if os.environ.get("SC_SCENE") == "street":
    from street_crafter_amd.scenes import make_street_scene
    scene = make_street_scene(N)[0].to(dev)
else:
    scene = make_scene(N).to(dev)
cam = make_camera(W, H, 2050.0 * W / 1920.0, 2050.0 * W / 1920.0).to(dev)
for t in (scene.means, scene.quats, scene.scales, scene.opacities, scene.sh):
    t.requires_grad_(True)
target = torch.rand(3, H, W, device=dev)


def step():
    for t in (scene.means, scene.quats, scene.scales, scene.opacities, scene.sh):
        t.grad = None
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    out = render_gaussians(scene, cam, mode="train")
    loss = (out["rgb"] - target).abs().mean() + 0.01 * out["acc"].mean()
    e1.record()
    loss.backward()
    e2.record()
    return e0, e1, e2, out


from street_crafter_amd import _lib  # noqa: E402
if os.environ.get("SC_RASTER_BWD"):
    _lib.set_option("raster_bwd", int(os.environ["SC_RASTER_BWD"]))
if os.environ.get("SC_TILE_ORDER"):
    from street_crafter_amd import rendering
    rendering.set_tile_order(bool(int(os.environ["SC_TILE_ORDER"])))
if os.environ.get("SC_BWD_SPLIT"):
    _lib.set_option("raster_bwd_split", int(os.environ["SC_BWD_SPLIT"]))
if os.environ.get("SC_RASTER_SPLIT"):
    _lib.set_option("raster_split", int(os.environ["SC_RASTER_SPLIT"]))
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
evs = [step() for _ in range(STEPS)]
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / STEPS
fwd = sorted(a.elapsed_time(b) for a, b, _, _ in evs)[STEPS // 2]
bwd = sorted(b.elapsed_time(c) for _, b, c, _ in evs)[STEPS // 2]
vp = evs[-1][3]["viewspace_points"]
print(f"N={N} {W}x{H}: wall {wall * 1e3:.3f} ms/step  fwd {fwd:.3f} ms  bwd {bwd:.3f} ms  "
      f"it/s {1.0 / wall:.1f}  absgrad set: {hasattr(vp, 'absgrad')}", flush=True)
