"""A/B timing of kernel variants / diagnostic skips on S-1M, interleaved in ONE process
(guide rule 24).  Prints per-operator HIP-event times per configuration.

    python tools/exp_kernels.py [n_gauss] [rounds] [street]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from street_crafter_amd import _lib, rendering  # noqa: E402
from harness.caller import render_gaussians  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene  # noqa: E402

_lib.use_diagnostic_build()            # debug0..3 exist in lib/libstreet_crafter_hip_diag.so only (build.py --diag)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
STREET = len(sys.argv) > 3 and sys.argv[3] == "street"
dev = "cuda"
scene = (make_street_scene(N)[0] if STREET else make_scene(N)).to(dev)
cam = make_camera().to(dev)

CONFIGS = [
    ("baseline", {}),
    ("scatter: no record store", {"debug0": 1}),
    ("scatter: no pass 2", {"debug0": 2}),
    ("raster: no blend loop", {"debug1": 1}),
    ("bucket sort: no final stores", {"debug2": 1}),
    ("bucket sort: no rank loop/stores", {"debug2": 2}),
    ("split: no copy pass", {"debug3": 4}),
    ("isect radix route", {"isect": "radix"}),
    ("raster variant 0", {"raster_fwd": 0}),
    ("raster variant 3", {"raster_fwd": 3}),
]


def run(cfg, frames=6):
    prev = {}
    for k, v in cfg.items():
        if k == "isect":
            prev[k] = rendering.set_isect_mode(v)
        else:
            prev[k] = _lib.set_option(k, v)
    ev = {}
    try:
        with torch.no_grad():
            for _ in range(2):
                render_gaussians(scene, cam)
            for _ in range(frames):
                render_gaussians(scene, cam, stage_events=ev)
        torch.cuda.synchronize()
    finally:
        for k, v in prev.items():
            if k == "isect":
                rendering.set_isect_mode(v)
            else:
                _lib.set_option(k, v)
    return {k: sorted(a.elapsed_time(b) for a, b in v)[len(v) // 2] for k, v in ev.items()}


res = {name: [] for name, _ in CONFIGS}
for r in range(ROUNDS):
    for name, cfg in CONFIGS:
        res[name].append(run(cfg))
for name, _ in CONFIGS:
    rs = res[name]
    keys = rs[0].keys()
    med = {k: sorted(x[k] for x in rs)[len(rs) // 2] for k in keys}
    print(f"{name:34s} " + "  ".join(f"{k[:10]}={v * 1e3:7.1f}us" for k, v in med.items()), flush=True)
