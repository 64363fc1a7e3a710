"""isect_tiles: the PULL route (sc_set_option "isect_pull" 1: buckets gather their own records) against the scatter route
(0) and the reference-shaped radix route, bit for bit, with per-route HIP-event times of the operator.
Usage: python tools/check_pull.py [scene ...]      scenes: s100k s1m street1m street3m sky small cams2 train"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from street_crafter_amd import _lib, rendering  # noqa: E402
import gsplat.rendering as R  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene  # noqa: E402

SCENES = {
    "s100k": lambda: (make_scene(100_000), 1920, 1280, 1),
    "s1m": lambda: (make_scene(1_000_000), 1920, 1280, 1),
    "train": lambda: (make_scene(1_000_000), 1600, 1066, 1),
    "street1m": lambda: (make_street_scene(1_000_000)[0], 1920, 1280, 1),
    "street3m": lambda: (make_street_scene(3_000_000)[0], 1920, 1280, 1),
    "sky": lambda: (make_street_scene(1_000_000)[1], 1920, 1280, 1),
    "small": lambda: (make_scene(20_000, seed=3, scale_range=(0.01, 0.6)), 333, 211, 1),
    "cams2": lambda: (make_scene(200_000, seed=5), 640, 400, 2),
}
which = sys.argv[1:] or ["s100k", "s1m", "street1m", "sky", "small", "cams2"]
bad = 0
for name in which:
    sc, W, H, C = SCENES[name]()
    sc = sc.to("cuda")
    tw, th = (W + 15) // 16, (H + 15) // 16
    for yaw in (0.0, 0.15):
        cams = [make_camera(W, H, 2050.0 * W / 1920.0, 2050.0 * W / 1920.0, yaw=yaw + 0.3 * c).to("cuda") for c in range(C)]
        vm = torch.stack([c.viewmat for c in cams])
        Ks = torch.stack([c.K for c in cams])
        with torch.no_grad():
            radii, m2, d, con, comp = R.fully_fused_projection(sc.means, None, sc.quats, sc.scales, vm, Ks, W, H,
                                                               near_plane=0.001, far_plane=1000.0, calc_compensations=True)
            out, times = {}, {}
            for route in ("pull", "scatter", "radix"):
                prev_mode = rendering.set_isect_mode("radix" if route == "radix" else "bin")
                prev_pull = _lib.set_option("isect_pull", 1 if route == "pull" else 0)
                rendering.reset_state()
                ts = []
                for rep in range(2 if route == "radix" else 8):        # (from the second call on: predicted sizes)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    tpg, ids, fids = R.isect_tiles(m2, radii, d, 16, tw, th, n_cameras=C)
                    off = R.isect_offset_encode(ids, C, tw, th)
                    n = fids.shape[0]
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                out[route] = (tpg.clone(), torch.as_tensor(ids).clone(), torch.as_tensor(fids).clone(), off.clone())
                times[route] = sorted(ts[1:])[len(ts[1:]) // 2] if len(ts) > 1 else ts[0]
                if route != "radix":
                    meta = list(rendering._STATE.last_meta.values())[-1]
                _lib.set_option("isect_pull", prev_pull)
                rendering.set_isect_mode(prev_mode)
        names = ("tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets")
        for route in ("pull", "scatter"):
            for nm, a, b in zip(names, out[route], out["radix"]):
                same = a.shape == b.shape and bool((a == b).all())
                if not same:
                    bad += 1
                    nd = int((a != b).sum()) if a.shape == b.shape else -1
                    print(f"{name} yaw {yaw}: {route} {nm} DIFFERENT from the radix route ({nd} of {a.numel()} elements)")
        print(f"{name} yaw {yaw}: I, records, largest bucket = {meta}; isect_tiles median us: "
              + "  ".join(f"{r} {t:.1f}" for r, t in times.items()), flush=True)
print("OK" if bad == 0 else f"FAILED: {bad} tensors differ")
sys.exit(1 if bad else 0)
