"""cProfile of the host side of the frame loop (caller sequence and fused path), S-1M."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from street_crafter_amd.dist import to_uint8_frame  # noqa: E402
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene  # noqa: E402
from gsplat.rendering import rasterization  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
scene = make_scene(N).to("cuda")
cam = make_camera().to("cuda")
op1 = scene.opacities[:, 0].contiguous()


def caller():
    with torch.no_grad():
        out = render_gaussians(scene, cam)
        return to_uint8_frame(out["rgb"])


def fused():
    with torch.no_grad():
        rc, _, _ = rasterization(scene.means, scene.quats, scene.scales, op1, scene.sh, cam.viewmat[None], cam.K[None],
                                 cam.width, cam.height, near_plane=cam.znear, far_plane=cam.zfar, sh_degree=scene.sh_degree,
                                 render_mode="RGB+ED", rasterize_mode="antialiased", camera_centers_=cam.camera_center[None])
        return to_uint8_frame(rc[0, ..., :3].permute(2, 0, 1))


for name, fn in (("caller sequence", caller), ("fused rasterization()", fused)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        fn()
    t1 = time.perf_counter()          # host time to ENQUEUE 200 frames (GPU may lag behind)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"== {name}: host enqueue {1e3 * (t1 - t0) / 200:.3f} ms/frame, wall {1e3 * (t2 - t0) / 200:.3f} ms/frame", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        fn()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("tottime").print_stats(14)
