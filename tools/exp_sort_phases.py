"""Prices the phases of super_sort_kernel / bin_scatter_flat_kernel with the DIAGNOSTIC build's skip switches: isect_tiles time by
HIP events with the kernel cut off after each phase (outputs invalid; every consumer bound-checks).
    python tools/exp_sort_phases.py [s1m|street] [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from street_crafter_amd import _lib, rendering
_lib.use_diagnostic_build()
import gsplat.rendering as R
from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene
which = sys.argv[1] if len(sys.argv) > 1 else "s1m"
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sc = (make_street_scene(1_000_000)[0] if which == "street" else make_scene(1_000_000)).to("cuda")
cam = make_camera().to("cuda")
rendering.set_deferred_isect(False)
with torch.no_grad():
    radii, m2, d, con, comp = R.fully_fused_projection(sc.means, None, sc.quats, sc.scales, cam.viewmat[None], cam.K[None], 1920, 1280,
                                                       near_plane=0.001, far_plane=1000.0, calc_compensations=True)
CONFIGS = [("full", {}), ("sort: stop after load + min/max", {"debug2": 4}), ("sort: ... + coarse count + scan", {"debug2": 8}),
           ("sort: ... + fine count + scan", {"debug2": 32}), ("sort: ... + LDS scatter", {"debug2": 2}),
           ("sort: ... + rank loop + emit pass A (no stores)", {"debug2": 1}),
           ("scatter: stop after load + tables", {"debug0": 4}), ("scatter: ... + count pass", {"debug0": 8}),
           ("scatter: ... + reservation", {"debug0": 2}), ("scatter: all but the record stores", {"debug0": 1})]
res = {n: [] for n, _ in CONFIGS}
for r in range(ROUNDS):
    for name, cfg in CONFIGS:
        prev = {k: _lib.set_option(k, v) for k, v in cfg.items()}
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            with torch.no_grad():
                R.isect_tiles(m2, radii, d, 16, 120, 80, n_cameras=1)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        for k, v in prev.items():
            _lib.set_option(k, v)
        res[name].append(sorted(ts[2:])[1])
for name, _ in CONFIGS:
    v = sorted(res[name])
    print(f"{name:52s} {v[len(v) // 2]:7.1f} us", flush=True)
