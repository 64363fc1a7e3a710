"""Renders N frames of one scene through the caller's sequence; meant to run under rocprofv3:
    rocprofv3 --kernel-trace --stats -d OUT -- python3 tools/prof_scene.py street1m 20
scenes: s1m | street1m | street3m | sky (the street scene's 31k-Gaussian sky set)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from street_crafter_amd.dist import to_uint8_frame  # noqa: E402
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "s1m"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if which == "s1m":
    sc = make_scene(1_000_000)
elif which == "street1m":
    sc = make_street_scene(1_000_000)[0]
elif which == "street3m":
    sc = make_street_scene(3_000_000)[0]
elif which == "sky":
    sc = make_street_scene(1_000_000)[1]
else:
    raise SystemExit(__doc__)
sc = sc.to("cuda")
for k in range(4):
    if os.environ.get(f"SC_DEBUG{k}"):
        from street_crafter_amd import _lib
        _lib.use_diagnostic_build()    # lib/libstreet_crafter_hip_diag.so (build.py --diag)
        _lib.set_option(f"debug{k}", int(os.environ[f"SC_DEBUG{k}"]))
cam = make_camera().to("cuda")
with torch.no_grad():
    for _ in range(frames):
        to_uint8_frame(render_gaussians(sc, cam)["rgb"])
torch.cuda.synchronize()
print("done", which, frames)
