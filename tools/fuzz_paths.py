"""Randomised cross-check of the fast paths against the reference-shaped ones on the GPU:
  * isect_tiles: tile-bucketed path vs count / emit / device radix sort  (bit-exact)
  * rasterize_to_pixels: wave-per-tile kernel (3) vs reference-shaped kernel (0)  (bit-exact)
  * rasterization(): fused forward vs composition  (bit-exact)
over random sizes, resolutions (ragged tiles), camera counts, depth ranges and splat sizes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gsplat.rendering as R  # noqa: E402
from street_crafter_amd import _lib, rendering  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
for it in range(ROUNDS):
    n = int(rng.choice([1, 7, 500, 20_000, 150_000, 600_000]))
    W = int(rng.integers(17, 4000)); H = int(rng.integers(17, 2300))
    C = int(rng.choice([1, 1, 2, 3]))
    if C * ((W + 15) // 16) * ((H + 15) // 16) > 36000:
        C = 1
    smax = float(rng.choice([0.02, 0.15, 0.6, 3.0]))
    zmin = float(rng.choice([0.5, 2.0, 10.0]))
    deg = int(rng.integers(0, 4))
    sc = make_scene(n, sh_degree=deg, seed=int(rng.integers(1 << 30)), z_range=(zmin, zmin * float(rng.choice([2, 40]))),
                    scale_range=(0.004, smax)).to("cuda")
    if rng.random() < 0.3 and n > 1000:        # street-shaped instead of i.i.d.: oversized super-tiles, huge splats
        fg, sky = make_street_scene(n, n_sky=max(8, n // 20), sh_degree=deg, seed=int(rng.integers(1 << 30)))
        sc = (fg if rng.random() < 0.7 else sky).to("cuda")
        n = sc.n
    if rng.random() < 0.3 and n > 10:          # many equal depths: only the id bits of the sort key separate them
        sc.means[:, 2] = torch.round(sc.means[:, 2])
    f = 2050.0 * W / 1920.0
    cams = [make_camera(W, H, f, f, yaw=0.05 * i, shift=(0.2 * i, 0.0, 0.0)) for i in range(C)]
    V = torch.stack([c.viewmat for c in cams]).cuda(); K = torch.stack([c.K for c in cams]).cuda()
    with torch.no_grad():
        radii, m2, d, con, comp = R.fully_fused_projection(sc.means, None, sc.quats, sc.scales, V, K, W, H,
                                                           near_plane=0.001, far_plane=1000.0, calc_compensations=True)
        tw, th = (W + 15) // 16, (H + 15) // 16
        outs = {}
        too_many = False
        for mode in ("bin", "radix"):
            prev = rendering.set_isect_mode(mode)
            try:
                tpg, ids, fids = R.isect_tiles(m2, radii, d, 16, tw, th, n_cameras=C)
                off = R.isect_offset_encode(ids, C, tw, th)
            except RuntimeError as e:          # big splats x many tiles: beyond int32 intersections, refused loudly
                if "exceed the int32 range" not in str(e):
                    raise
                too_many = True
                break
            finally:
                rendering.set_isect_mode(prev)
            outs[mode] = (tpg, ids, fids, off)
        if too_many:
            print(f"[{it}] N={n} C={C} {W}x{H} smax={smax} z>={zmin}: more than 2^31 - 1 intersections, refused by both routes", flush=True)
            continue
        ok_isect = all(torch.equal(a, b) for a, b in zip(outs["bin"], outs["radix"]))
        op = sc.opacities[None, :, 0] * comp
        cols = torch.rand(C, n, 4, device="cuda")
        imgs = {}
        for v in (3, 0):
            prev = _lib.set_option("raster_fwd", v)
            try:
                imgs[v] = R.rasterize_to_pixels(m2, con, cols, op, W, H, 16, outs["bin"][3], outs["bin"][2])
            finally:
                _lib.set_option("raster_fwd", prev)
        ok_raster = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(imgs[3], imgs[0]))
        fr = []
        for k, fused in enumerate((True, True, False)):   # the fused path twice: cold, then with a warm dispatch-list hint
            if k == 1:
                fr.pop()
            prev = rendering.set_fused_rasterization(fused)
            try:
                fr.append(R.rasterization(sc.means, sc.quats, sc.scales, sc.opacities[:, 0], sc.sh, V, K, W, H,
                                          near_plane=0.001, far_plane=1000.0, sh_degree=deg, render_mode="RGB+ED",
                                          rasterize_mode="antialiased"))
            finally:
                rendering.set_fused_rasterization(prev)
        ok_fused = torch.equal(fr[0][0].view(torch.int32), fr[1][0].view(torch.int32)) and \
            torch.equal(fr[0][1].view(torch.int32), fr[1][1].view(torch.int32)) and fr[0][2]["fused"] and not fr[1][2]["fused"]
    I = outs["bin"][1].numel()
    print(f"[{it}] N={n} C={C} {W}x{H} smax={smax} z>={zmin} deg={deg} I={I}: isect {'ok' if ok_isect else 'MISMATCH'}, "
          f"raster {'ok' if ok_raster else 'MISMATCH'}, fused {'ok' if ok_fused else 'MISMATCH'}", flush=True)
    bad += (not ok_isect) + (not ok_raster) + (not ok_fused)
print("FAILED" if bad else "all paths agree")
sys.exit(1 if bad else 0)
