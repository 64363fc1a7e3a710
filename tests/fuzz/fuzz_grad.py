"""Randomised check of the rasterizer's BACKWARD (both kernels) against the float64 autograd oracle
(oracle/gsplat_torch.py) on small random scenes: gradients of means2d, conics, colours, opacities and the exact
absgrad, 2e-3 of the largest entry (the bar of tests/test_gpu_parity.py), pixels the oracle flags threshold-unstable
left out of the loss.  Test infrastructure (it uses oracle/): lives under tests/, never imported by the product.
Usage: python tests/fuzz/fuzz_grad.py [seed] [rounds]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gsplat.rendering as R  # noqa: E402
from oracle import gsplat_oracle as O  # noqa: E402
from oracle import gsplat_torch as OT  # noqa: E402
from street_crafter_amd import _lib  # noqa: E402
from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
DEV = "cuda"
bad = 0


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


for it in range(ROUNDS):
    n = int(rng.choice([1, 20, 300, 1500]))
    W = int(rng.integers(20, 200)); H = int(rng.integers(20, 140))
    D = int(rng.choice([3, 4, 4, 6]))
    use_bg = bool(rng.random() < 0.5)
    smax = float(rng.choice([0.05, 0.3, 1.0]))
    sc = make_scene(n, sh_degree=1, seed=int(rng.integers(1 << 30)), z_range=(1.0, float(rng.choice([4.0, 40.0]))),
                    scale_range=(0.01, smax))
    if n >= 300 and rng.random() < 0.4:
        sc = make_street_scene(n, seed=int(rng.integers(1 << 30)))[0]
    f = 1.1 * W * float(rng.choice([0.6, 1.0, 2.5]))
    cam = make_camera(W, H, f, f, yaw=float(rng.choice([0.0, 0.3])))
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    with torch.no_grad():
        radii, m2, d, con, comp = R.fully_fused_projection(sc.means.to(DEV), None, sc.quats.to(DEV), sc.scales.to(DEV),
                                                           cam.viewmat[None].to(DEV), cam.K[None].to(DEV), W, H,
                                                           near_plane=0.01, far_plane=1000.0, calc_compensations=True)
        tpg, ids, fids = R.isect_tiles(m2, radii, d, 16, tw, th, n_cameras=1)
        offs = R.isect_offset_encode(ids, 1, tw, th)
    if fids.numel() == 0:
        print(f"[{it}] N={sc.n} {W}x{H}: nothing visible", flush=True)
        continue
    op = (sc.opacities[None, :, 0].to(DEV) * comp).cpu().numpy()
    src = (m2.cpu().numpy(), con.cpu().numpy(), rng.uniform(0, 1, size=(1, sc.n, D)).astype(np.float32), op)
    bg = rng.uniform(0, 1, size=(1, D)).astype(np.float32) if use_bg else None
    offs_np, fids_np = offs.cpu().numpy(), fids.cpu().numpy()
    w_c = rng.normal(size=(1, H, W, D)).astype(np.float32)
    w_a = rng.normal(size=(1, H, W, 1)).astype(np.float32)
    _, _, _, unstable = O.rasterize_to_pixels(src[0], src[1], src[2], src[3], W, H, 16, offs_np, fids_np, return_unstable=True)
    w_c[unstable] = 0.0
    w_a[unstable] = 0.0
    ref = [torch.from_numpy(a).double().requires_grad_(True) for a in src]
    pix = []
    rcr, rar = OT.rasterize_to_pixels(ref[0], ref[1], ref[2], ref[3], W, H, 16, torch.from_numpy(offs_np), torch.from_numpy(fids_np),
                                      backgrounds=None if bg is None else torch.from_numpy(bg).double(), pixel_grads=pix)
    ((rcr * torch.from_numpy(w_c).double()).sum() + (rar * torch.from_numpy(w_a).double()).sum()).backward()
    ref_abs = OT.absgrad_from_pixel_grads(pix, sc.n).numpy()
    worst = {}
    for variant in (1, 0):
        hip = [torch.from_numpy(a).to(DEV).requires_grad_(True) for a in src]
        rc, ra = R.rasterize_to_pixels(hip[0], hip[1], hip[2], hip[3], W, H, 16, offs, fids,
                                       backgrounds=None if bg is None else torch.from_numpy(bg).to(DEV), absgrad=True)
        prev = _lib.set_option("raster_bwd", variant)
        try:
            ((rc * torch.from_numpy(w_c).to(DEV)).sum() + (ra * torch.from_numpy(w_a).to(DEV)).sum()).backward()
        finally:
            _lib.set_option("raster_bwd", prev)
        for h_, r_, name in zip(hip, ref, ("means2d", "conics", "colors", "opacities")):
            g_ = r_.grad.numpy() if r_.grad is not None else np.zeros_like(src[0])
            if np.abs(g_).max() > 0:
                worst[(variant, name)] = rel(h_.grad.cpu().numpy(), g_)
        if np.abs(ref_abs).max() > 0:
            worst[(variant, "absgrad")] = rel(hip[0].absgrad.cpu().numpy()[0], ref_abs)
    w = max(worst.values()) if worst else 0.0
    ok = w < 2e-3
    print(f"[{it}] N={sc.n} {W}x{H} D={D} bg={use_bg} smax={smax} f={f:.0f} I={fids_np.size} unstable={int(unstable.sum())}: "
          f"worst {w:.2e} {max(worst, key=worst.get) if worst else ''}: {'ok' if ok else 'FAIL'}", flush=True)
    bad += not ok
print("FAILED" if bad else "both backward kernels agree with the float64 oracle")
