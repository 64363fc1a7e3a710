"""Stage times of the order job (the one workgroup that builds the rasterizer's dispatch list), from the device's
100-MHz clock: debug1 bit 22.  Usage: python tools/exp_order_job.py [s1m|sky|street1m]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from street_crafter_amd import _lib  # noqa: E402
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene  # noqa: E402

_lib.use_diagnostic_build()            # the stage stamps exist in the diagnostic build only (build.py --diag)
which = sys.argv[1] if len(sys.argv) > 1 else "s1m"
sc = {"s1m": lambda: make_scene(1_000_000), "street1m": lambda: make_street_scene(1_000_000)[0],
      "sky": lambda: make_street_scene(1_000_000)[1]}[which]().to("cuda")
cam = make_camera().to("cuda")
T = 120 * 80
n_fwd = T + T // 8 + 8
with torch.no_grad():
    for _ in range(4):
        render_gaussians(sc, cam)
    _lib.set_option("debug1", 64 << 16)
    rows = []
    for _ in range(6):
        o = render_gaussians(sc, cam, return_intermediates=True)
        torch.cuda.synchronize()
        rows.append(o["_isect_offsets"]._sc_sched[0][n_fwd - 8: n_fwd - 1].cpu().tolist())
    _lib.set_option("debug1", 0)
names = ["start", "snapshot+max", "horizontal max", "class sweep", "scan+split", "scatter sweep", "write-out"]
for r in rows:
    print(which, " ".join(f"{n} {v / 100:.1f}us" for n, v in zip(names, r)))
