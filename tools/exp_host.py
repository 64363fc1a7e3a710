"""Host-side (Python) cost of one frame: run the pipeline on a tiny scene so the GPU is never the
bottleneck, report wall time per frame and a cProfile of where the host time goes."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from street_crafter_amd.dist import to_uint8_frame  # noqa: E402
from street_crafter_amd.pipeline import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene  # noqa: E402

dev = "cuda"
scene = make_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 2000).to(dev)
cam = make_camera(256, 160, 280.0, 280.0).to(dev)


def frames(n, ev=None):
    with torch.no_grad():
        for _ in range(n):
            out = render_gaussians(scene, cam, stage_events=ev)
            to_uint8_frame(out["rgb"])


frames(20)
torch.cuda.synchronize()
for label, ev in (("no events", None), ("with stage events", {})):
    t0 = time.perf_counter()
    frames(300, ev)
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us/frame host-bound wall")
pr = cProfile.Profile()
pr.enable()
frames(300)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
