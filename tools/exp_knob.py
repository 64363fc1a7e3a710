"""Renders a few S-1M frames with sc_set_option knobs given as KEY=VALUE arguments (for rocprofv3
kernel-trace A/B of a single kernel under diagnostic skips).

    rocprofv3 --kernel-trace --stats ... -- python3 tools/exp_knob.py debug0=1
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from street_crafter_amd import _lib  # noqa: E402
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene  # noqa: E402

scene = make_scene(1_000_000).to("cuda")
cam = make_camera().to("cuda")
if any(kv.startswith("debug") for kv in sys.argv[1:]):
    _lib.use_diagnostic_build()        # the skip switches exist in lib/libstreet_crafter_hip_diag.so only
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    _lib.set_option(k, int(v))
with torch.no_grad():
    for _ in range(8):
        render_gaussians(scene, cam)
torch.cuda.synchronize()
