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

if os.environ.get("FUZZ_LIB"):       # FUZZ_LIB=tag: check an experiment build (SC_DIAG_TAG=tag ... build --diag) instead of the shipped library
    _lib.use_diagnostic_build("" if os.environ["FUZZ_LIB"] == "diag" else os.environ["FUZZ_LIB"])
    _lib.set_fast_binding(False)     # (the compiled binding layer is linked to the shipped library)
if os.environ.get("FUZZ_PULL"):      # FUZZ_PULL=1: the bucketed route under test is the PULL route (sc_set_option "isect_pull")
    _lib.set_option("isect_pull", 1)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
ONLY = int(os.environ.get("FUZZ_ONLY", -1))
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
    # focal length and view direction vary too: a long lens turns every splat into a big one, a turned camera puts the
    # scene's edge -- and with it splats clipped by the frame border -- into view
    f = 2050.0 * W / 1920.0 * float(rng.choice([0.5, 1.0, 1.0, 3.0]))
    yaw0 = float(rng.choice([0.0, 0.0, 0.25, -0.4]))
    lift = float(rng.choice([0.0, 0.0, 1.5]))
    cams = [make_camera(W, H, f, f, yaw=yaw0 + 0.05 * i, shift=(0.2 * i, lift, 0.0)) for i in range(C)]
    if ONLY >= 0 and it != ONLY:           # FUZZ_ONLY=k: replay iteration k of this seed alone (all draws are made above)
        continue
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
        if not ok_isect:
            print("    bucketed route meta (n_isects, n_records, largest super-tile):", list(rendering._STATE.last_meta.values())[-1:])
            offb = outs["radix"][3].reshape(-1).long()
            dd = (outs["bin"][2] != outs["radix"][2]).nonzero().reshape(-1)
            tiles_bad = torch.unique(torch.searchsorted(offb, dd, right=True) - 1)
            tw_ = (W + 15) // 16
            print("    tiles with a wrong list:", tiles_bad.numel(), "first (tx,ty):", [(int(t) % tw_, int(t) // tw_) for t in tiles_bad[:12]],
                  "list lengths:", [int((offb[t + 1] if t + 1 < offb.numel() else outs['radix'][2].numel()) - offb[t]) for t in tiles_bad[:12]])
            for mode_ in ("bin", "radix"):          # (no indexing with the lists' own ids: they may be garbage)
                ids_, fids_ = outs[mode_][1], outs[mode_][2]
                print(f"    {mode_}: isect_ids ascending {bool((ids_[1:] >= ids_[:-1]).all())}, flatten_ids in [{int(fids_.min())}, "
                      f"{int(fids_.max())}] of {C * n}, sum {int(fids_.long().sum())}")
            for name, a, b in zip(("tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets"), outs["bin"], outs["radix"]):
                if a.shape != b.shape:
                    print(f"    {name}: shapes {tuple(a.shape)} vs {tuple(b.shape)}")
                elif not torch.equal(a, b):
                    d = (a.reshape(-1) != b.reshape(-1)).nonzero().reshape(-1)
                    print(f"    {name}: {d.numel()} of {a.numel()} differ, first at {int(d[0])}: bin {a.reshape(-1)[d[:4]].tolist()} "
                          f"radix {b.reshape(-1)[d[:4]].tolist()}, last at {int(d[-1])}")
        op = sc.opacities[None, :, 0] * comp
        D = 3 if (it % 3 == 1) else 4                       # the wave kernel exists for 3 and 4 channels
        cols = torch.rand(C, n, D, device="cuda")
        bgs = torch.rand(C, D, device="cuda") if it % 2 == 0 else None
        tmask = (torch.rand(C, th, tw, device="cuda") < 0.7) if it % 4 == 3 else None
        imgs = {}
        for v in (3, 0):
            prev = _lib.set_option("raster_fwd", v)
            try:
                imgs[v] = R.rasterize_to_pixels(m2, con, cols, op, W, H, 16, outs["bin"][3], outs["bin"][2],
                                                backgrounds=bgs, masks=tmask)
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
    # backward: the wave-per-tile kernel (with the dispatch list of the second, warm call: half tiles included) against
    # the reference-shaped one, through the train-mode render of the first camera
    ok_bwd = True
    if n <= 150_000 and W * H <= 1600 * 1100:
        from harness.caller import render_gaussians
        cam0 = cams[0].to("cuda")
        target = torch.rand(3, H, W, device="cuda")
        grads = []
        for variant in (1, 1, 0):
            params = [t.detach().clone().requires_grad_(True) for t in (sc.means, sc.quats, sc.scales, sc.opacities, sc.sh)]
            scn = type(sc)(*params, sc.sh_degree)
            prev = _lib.set_option("raster_bwd", variant)
            try:
                out = render_gaussians(scn, cam0, mode="train")
                ((out["rgb"] - target).abs().mean() + 0.05 * out["acc"].mean() + 0.01 * out["depth"].mean()).backward()
            finally:
                _lib.set_option("raster_bwd", prev)
            vp = out["viewspace_points"]
            grads.append([params[3].grad.detach().clone(), params[4].grad.detach().clone(), vp.grad.detach().clone(),
                          vp.absgrad.detach().clone()])
        # (the rasterizer's own outputs and the well-conditioned parameter gradients: through the projection's VJP the
        # float-atomic noise of EITHER kernel reaches 5e-4 .. 2e-2 of the largest quaternion gradient for splats metres
        # wide -- two runs of the reference-shaped kernel differ by as much, tools/debug_bwd_fuzz.py)
        for name, a, b in zip(("opacities", "sh", "means2d", "absgrad"), grads[1], grads[2]):
            # (means2d: pixel contributions of both signs nearly cancel for an isolated splat; its scale is absgrad's)
            den = float((grads[2][3] if name == "means2d" else b).abs().max()) + 1e-20
            err = float((a - b).abs().max()) / den
            if not bool(torch.isfinite(a).all()) or err > 5e-4:
                ok_bwd = False
                print(f"    backward: {name} differs by {err:.3e} of its largest entry ({den:.3e}); finite {bool(torch.isfinite(a).all())}")
    I = outs["bin"][1].numel()
    print(f"[{it}] N={n} C={C} {W}x{H} smax={smax} z>={zmin} deg={deg} I={I}: isect {'ok' if ok_isect else 'MISMATCH'}, "
          f"raster {'ok' if ok_raster else 'MISMATCH'}, fused {'ok' if ok_fused else 'MISMATCH'}, "
          f"backward {'ok' if ok_bwd else 'MISMATCH'}", flush=True)
    bad += (not ok_isect) + (not ok_raster) + (not ok_fused) + (not ok_bwd)
print("FAILED" if bad else "all paths agree")
sys.exit(1 if bad else 0)
