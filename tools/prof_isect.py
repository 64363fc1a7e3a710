"""isect_tiles alone, N calls of one route, for rocprofv3 --kernel-trace --stats.
Usage: python tools/prof_isect.py [s1m|street1m|...] [pull|scatter] [calls]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from street_crafter_amd import _lib, rendering  # noqa: E402
import gsplat.rendering as R  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "s1m"
route = sys.argv[2] if len(sys.argv) > 2 else "pull"
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 20
sc = {"s1m": lambda: make_scene(1_000_000), "s100k": lambda: make_scene(100_000),
      "street1m": lambda: make_street_scene(1_000_000)[0], "street3m": lambda: make_street_scene(3_000_000)[0],
      "sky": lambda: make_street_scene(1_000_000)[1]}[which]().to("cuda")
W, H = 1920, 1280
cam = make_camera(W, H).to("cuda")
_lib.set_option("isect_pull", 1 if route == "pull" else 0)
with torch.no_grad():
    radii, m2, d, con, comp = R.fully_fused_projection(sc.means, None, sc.quats, sc.scales, cam.viewmat[None], cam.K[None], W, H,
                                                       near_plane=0.001, far_plane=1000.0, calc_compensations=True)
    for _ in range(calls):
        tpg, ids, fids = R.isect_tiles(m2, radii, d, 16, 120, 80, n_cameras=1)
        n = fids.shape[0]
        torch.cuda.synchronize()
print(route, which, "I =", n)
