"""isect_tiles of one big scene: tile-bucketed route against the reference-shaped count / emit / radix-sort route,
bit for bit, optionally with an experiment build.
Usage: [FUZZ_LIB=tag] python tools/check_bin_vs_radix.py [street1m|street3m|s1m|sky] [W H]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from street_crafter_amd import _lib, rendering  # noqa: E402

if os.environ.get("FUZZ_LIB"):
    _lib.use_diagnostic_build("" if os.environ["FUZZ_LIB"] == "diag" else os.environ["FUZZ_LIB"])
    _lib.set_fast_binding(False)
import gsplat.rendering as R  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "street1m"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1280)
sc = {"s1m": lambda: make_scene(1_000_000), "street1m": lambda: make_street_scene(1_000_000)[0],
      "street3m": lambda: make_street_scene(3_000_000)[0], "sky": lambda: make_street_scene(1_000_000)[1]}[which]().to("cuda")
bad = 0
for yaw in (0.0, 0.2):
    cam = make_camera(W, H, 2050.0 * W / 1920.0, 2050.0 * W / 1920.0, yaw=yaw).to("cuda")
    with torch.no_grad():
        radii, m2, d, con, comp = R.fully_fused_projection(sc.means, None, sc.quats, sc.scales, cam.viewmat[None], cam.K[None],
                                                           W, H, near_plane=0.001, far_plane=1000.0, calc_compensations=True)
        tw, th = (W + 15) // 16, (H + 15) // 16
        out = {}
        for mode in ("bin", "radix"):
            prev = rendering.set_isect_mode(mode)
            for rep in range(2):        # (second call: predicted sizes, speculative launch)
                tpg, ids, fids = R.isect_tiles(m2, radii, d, 16, tw, th, n_cameras=1)
                off = R.isect_offset_encode(ids, 1, tw, th)
            out[mode] = (tpg.clone(), torch.as_tensor(ids).clone(), fids.clone(), off.clone())
            rendering.set_isect_mode(prev)
        names = ("tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets")
        for nm, a, b in zip(names, out["bin"], out["radix"]):
            same = a.shape == b.shape and bool((a == b).all())
            bad += 0 if same else 1
            print(f"{which} yaw {yaw}: {nm} {'equal' if same else 'DIFFERENT'} ({a.numel()} elements)")
        key = [k for k in rendering._STATE.last_meta][-1]
        print("  (n_isects, n_records, largest bucket) =", rendering._STATE.last_meta[key])
print("OK" if bad == 0 else f"FAILED: {bad} tensors differ")
sys.exit(1 if bad else 0)
