"""Renders N frames of S-<n> through the caller's sequence, one frame in flight, synchronising after every frame
(so that every frame starts on an idle GPU, as in a host-bound loop); meant for rocprofv3 --kernel-trace +
tools/trace_sequence.py: where does the latency of a SMALL frame go?
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/prof_small.py 100000 12"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.dist import to_uint8_frame  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 12
sync = len(sys.argv) <= 3 or sys.argv[3] != "nosync"
sc = make_scene(n).to("cuda")
cam = make_camera().to("cuda")
with torch.no_grad():
    for _ in range(frames):
        to_uint8_frame(render_gaussians(sc, cam)["rgb"])
        if sync:
            torch.cuda.synchronize()
torch.cuda.synchronize()
print("done", n, frames)
