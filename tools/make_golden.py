"""Generates the committed fixtures under tests/golden/.  Run in the BUILD container only:

    python tools/make_golden.py

Two kinds of vectors:
 * *_ref.npz  -- outputs of the REFERENCE's own Python code, imported from /root/reference
                 (street_gaussian/utils/sh_utils.py: eval_sh; street_gaussian/utils/loss_utils.py:
                 psnr).  These pin the oracle against reference code.  Only data is stored.
 * *_small.npz / knn_*.npz -- inputs + outputs of this repository's oracle (oracle/), stored so
                 that (a) the oracle cannot drift silently and (b) GPU tests have fixed vectors.
                 They do NOT pin anything against the reference (gsplat / simple-knn sources are
                 not vendored there): "parity unpinned", see oracle/gsplat_oracle.py.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def ref_vectors():
    sys.path.insert(0, REF)
    from street_gaussian.utils.sh_utils import eval_sh, IDFT    # noqa: E402
    from street_gaussian.utils.loss_utils import psnr           # noqa: E402
    # the actors' Fourier colour basis (consumed by street_crafter_amd/scene_io.py)
    times = torch.tensor([0.0, 0.125, 0.5, 0.77, 1.0, 2.5])
    np.savez_compressed(os.path.join(GOLD, "idft_ref.npz"), times=times.numpy(),
                        dim5=IDFT(times, 5).numpy(), dim8=IDFT(times, 8).numpy(), dim1=IDFT(times, 1).numpy())
    g = torch.Generator().manual_seed(11)
    n = 512
    d = torch.randn(n, 3, generator=g, dtype=torch.float64)
    d = d / d.norm(dim=-1, keepdim=True)
    coeffs = torch.randn(n, 25, 3, generator=g, dtype=torch.float64)       # [N,K,3] (gsplat layout)
    out = {"dirs": d.numpy(), "coeffs": coeffs.numpy()}
    for deg in range(5):
        # reference layout is [..., C, K]
        out[f"deg{deg}"] = eval_sh(deg, coeffs.permute(0, 2, 1), d).numpy()
    np.savez_compressed(os.path.join(GOLD, "sh_eval_ref.npz"), **out)

    a = torch.rand(3, 24, 32, generator=g)
    b = (a + 0.05 * torch.randn(3, 24, 32, generator=g)).clamp(0, 1)
    np.savez_compressed(os.path.join(GOLD, "psnr_ref.npz"), img1=a.numpy(), img2=b.numpy(),
                        psnr=np.float64(psnr(a, b).item()))
    sys.path.remove(REF)


def oracle_vectors():
    from oracle import gsplat_oracle as O
    from oracle import knn_oracle as KO
    from street_crafter_amd.scenes import make_camera, make_edge_case_scene, make_scene

    # (1) projection edge cases, N = 4096, full-size camera
    sc = make_edge_case_scene(4096)
    cam = make_camera()
    radii, m2, dep, con, comp = O.fully_fused_projection(
        sc.means.numpy(), sc.quats.numpy(), sc.scales.numpy(), cam.viewmat.numpy(), cam.K.numpy(),
        cam.width, cam.height, near_plane=cam.znear, far_plane=cam.zfar)
    np.savez_compressed(os.path.join(GOLD, "proj_small.npz"), means=sc.means.numpy(), quats=sc.quats.numpy(),
                        scales=sc.scales.numpy(), viewmat=cam.viewmat.numpy(), K=cam.K.numpy(),
                        width=cam.width, height=cam.height, near=cam.znear, far=cam.zfar, radii=radii,
                        means2d=m2, depths=dep, conics=con, compensations=comp)

    projection_variant_vectors()

    # (2)+(3) small full pipeline: 128 x 96 image (8 x 6 tiles), 3000 Gaussians, 4 channels
    _oracle_vectors_rest(O, KO, make_camera, make_scene)


def projection_variant_vectors():
    """(1b) the projection under the upstream-version-dependent constants (oracle/gsplat_oracle.py
    PROJ_CLAMPS / RADIUS_FLOORS): edge cases seen through a camera whose principal point is OFF centre (the two clamps
    agree for a centred one); outputs per variant -> tests/golden/proj_variants.npz"""
    from oracle import gsplat_oracle as O
    sc, cam = projection_variant_case()
    var = {}
    for clamp in O.PROJ_CLAMPS:
        for floor in O.RADIUS_FLOORS:
            r, m2, dep, con, comp = O.fully_fused_projection(
                sc.means.numpy(), sc.quats.numpy(), sc.scales.numpy(), cam.viewmat.numpy(), cam.K.numpy(), cam.width,
                cam.height, near_plane=cam.znear, far_plane=cam.zfar, proj_clamp=clamp, radius_floor=floor)
            tag = f"{clamp}_{floor}"
            var.update({f"{tag}_radii": r, f"{tag}_means2d": m2, f"{tag}_depths": dep, f"{tag}_conics": con,
                        f"{tag}_compensations": comp})
    assert not np.array_equal(var["symmetric_0.01_conics"], var["asymmetric_0.01_conics"])
    assert not np.array_equal(var["symmetric_0.01_radii"], var["symmetric_0.1_radii"])
    # (inputs stored as well: torch's CPU normal generator is not bit-identical across hosts)
    np.savez_compressed(os.path.join(GOLD, "proj_variants.npz"), means=sc.means.numpy(), quats=sc.quats.numpy(),
                        scales=sc.scales.numpy(), viewmat=cam.viewmat.numpy(), K=cam.K.numpy(), width=cam.width,
                        height=cam.height, near=cam.znear, far=cam.zfar, **var)


def _oracle_vectors_rest(O, KO, make_camera, make_scene):
    cam = make_camera(width=128, height=96, fx=140.0, fy=140.0)
    sc = make_scene(3000, sh_degree=1, seed=5, x_span=0.6, y_span=0.45, z_range=(1.0, 30.0),
                    scale_range=(0.01, 0.4))
    r = O.render_frame(sc.means.numpy(), sc.quats.numpy(), sc.scales.numpy(), sc.opacities.numpy(),
                       sc.sh.numpy(), cam.viewmat.numpy(), cam.K.numpy(), cam.width, cam.height, 1,
                       near_plane=cam.znear, far_plane=cam.zfar, return_unstable=True)
    np.savez_compressed(os.path.join(GOLD, "pipeline_small.npz"), in_means=sc.means.numpy(),
                        in_quats=sc.quats.numpy(), in_scales=sc.scales.numpy(),
                        in_opacities=sc.opacities.numpy(), in_sh=sc.sh.numpy(), in_viewmat=cam.viewmat.numpy(),
                        in_K=cam.K.numpy(), in_width=cam.width, in_height=cam.height, in_near=cam.znear,
                        in_far=cam.zfar, **{k: v for k, v in r.items()})

    # (5) knn
    g = np.random.default_rng(3)
    pts = g.normal(size=(1000, 3)).astype(np.float32)
    dup = pts.copy()
    dup[100:140] = dup[100]                   # 40 exact duplicates
    line = np.zeros((257, 3), np.float32)
    line[:, 0] = np.arange(257, dtype=np.float32) * 0.25
    np.savez_compressed(os.path.join(GOLD, "knn_small.npz"), pts=pts, out=KO.dist_cuda2(pts), dup=dup,
                        out_dup=KO.dist_cuda2(dup), line=line, out_line=KO.dist_cuda2(line),
                        tiny=pts[:3], out_tiny=KO.dist_cuda2(pts[:3]))


def projection_variant_case():
    """Inputs of tests/golden/proj_variants.npz."""
    from street_crafter_amd.scenes import make_camera, make_edge_case_scene
    sc = make_edge_case_scene(1536, seed=23)
    cam = make_camera(960, 600, 520.0, 540.0, yaw=0.05, shift=(0.1, 0.05, -0.2))
    cam.K[0, 2] = 0.42 * 960          # principal point off centre: the asymmetric clamp's limits differ left / right
    cam.K[1, 2] = 0.57 * 600
    return sc, cam


PROJ_GROUPS = ("means2d_x", "means2d_y", "depths", "conic_a", "conic_b", "conic_c", "compensation")


def projection_bwd_case(which):
    """Inputs of the per-output projection-backward fixture: 'plain' = an ordinary small scene, 'clamped' =
    Gaussians far off to the side of the frustum but wide enough to reach the image, so that the EWA Jacobian
    uses the CLAMPED x/z, y/z (the clamp's derivative is zero there)."""
    import torch
    from street_crafter_amd.scenes import make_camera, make_scene
    cam = make_camera(320, 200, 350.0, 350.0, yaw=0.07, shift=(0.2, -0.1, 0.3))
    if which == "plain":
        sc = make_scene(360, seed=13, z_range=(1.0, 40.0), scale_range=(0.002, 0.5))    # needles included
        return sc.means, sc.quats, sc.scales, cam
    sc = make_scene(360, seed=14, z_range=(2.0, 20.0), scale_range=(0.3, 1.5))
    m = sc.means.clone()
    g = torch.Generator().manual_seed(3)
    lim = 1.3 * 0.5 * 320 / 350.0                       # tan-space clamp limit in x
    side = torch.where(torch.rand(360, generator=g) < 0.5, -1.0, 1.0)
    m[:, 0] = m[:, 2] * side * (lim + 0.02 + 0.25 * torch.rand(360, generator=g))
    limy = 1.3 * 0.5 * 200 / 350.0
    m[::3, 1] = m[::3, 2] * (limy + 0.02 + 0.1 * torch.rand(120, generator=g))
    return m.contiguous(), sc.quats, sc.scales, cam


def raster_bwd_weights(unstable, seed, W=128, H=96, D=4):
    """Upstream gradients of the rasterize-backward fixture; threshold-unstable pixels take no part in the loss."""
    rng = np.random.default_rng(seed)
    w_c = rng.normal(size=(1, H, W, D)).astype(np.float32)
    w_a = rng.normal(size=(1, H, W, 1)).astype(np.float32)
    w_c[unstable] = 0.0
    w_a[unstable] = 0.0
    return w_c, w_a


def backward_vectors():
    """gradients of the float64 autograd oracle (oracle/gsplat_torch.py), committed so that
    (a) drift of the gradient oracle is caught on the CPU and (b) the GPU tests need not recompute it."""
    import torch
    from oracle import gsplat_oracle as O
    from oracle import gsplat_torch as OT
    out = {}
    # ---- rasterize backward on pipeline_small (weights regenerated from the stored seed) ----
    g = np.load(os.path.join(GOLD, "pipeline_small.npz"))
    N, W, H, D = g["means2d"].shape[0], 128, 96, 4
    w_c, w_a = raster_bwd_weights(g["unstable"], 2024)
    src = (g["means2d"][None], g["conics"][None], g["colors"][None], g["opacities"][None])
    ref = [torch.from_numpy(a).double().requires_grad_(True) for a in src]
    pix = []
    rc, ra = OT.rasterize_to_pixels(ref[0], ref[1], ref[2], ref[3], W, H, 16, torch.from_numpy(g["isect_offsets"]),
                                    torch.from_numpy(g["flatten_ids"]), pixel_grads=pix)
    ((rc * torch.from_numpy(w_c).double()).sum() + (ra * torch.from_numpy(w_a).double()).sum()).backward()
    out.update(raster_seed=2024,          # (the weights are regenerated from the seed: raster_bwd_weights())
               raster_v_means2d=ref[0].grad.numpy(), raster_v_conics=ref[1].grad.numpy(),
               raster_v_colors=ref[2].grad.numpy(), raster_v_opacities=ref[3].grad.numpy(),
               raster_absgrad=OT.absgrad_from_pixel_grads(pix, N).numpy())
    # ---- spherical harmonics backward, degrees 0..4 ----
    rng = np.random.default_rng(77)
    dirs = rng.normal(size=(120, 3))
    dirs *= rng.uniform(0.2, 30.0, size=(120, 1))        # not normalised, as the caller passes them
    coeffs = rng.normal(size=(120, 25, 3))
    v_col = rng.normal(size=(120, 3))
    out.update(sh_dirs=dirs.astype(np.float32), sh_coeffs=coeffs.astype(np.float32), sh_v_colors=v_col.astype(np.float32))
    for deg in range(5):
        K = (deg + 1) ** 2
        d = torch.from_numpy(out["sh_dirs"]).double().requires_grad_(True)
        c = torch.from_numpy(out["sh_coeffs"][:, :K]).double().requires_grad_(True)
        (OT.spherical_harmonics(deg, d, c) * torch.from_numpy(out["sh_v_colors"]).double()).sum().backward()
        out[f"sh_v_coeffs_deg{deg}"] = c.grad.numpy()
        out[f"sh_v_dirs_deg{deg}"] = d.grad.numpy() if d.grad is not None else np.zeros((120, 3))   # degree 0: constant
    # ---- projection backward, ONE output at a time (unit upstream gradient on that output only) ----
    for which in ("plain", "clamped"):
        means, quats, scales, cam = projection_bwd_case(which)
        V, K = cam.viewmat.double(), cam.K.double()
        for gi, name in enumerate(PROJ_GROUPS):
            ref = [t.clone().double().requires_grad_(True) for t in (means, quats, scales)]
            radii, m2, dep, con, comp = OT.fully_fused_projection(ref[0], ref[1], ref[2], V, K, 320, 200,
                                                                  near_plane=0.001, far_plane=1000.0)
            outs = (m2[:, 0], m2[:, 1], dep, con[:, 0], con[:, 1], con[:, 2], comp)
            outs[gi].sum().backward()
            # [N, 10] = d out / d (means 3, quats 4, scales 3); an input an output does not depend on has no grad
            out[f"proj_{which}_{name}"] = np.concatenate(
                [r.grad.numpy() if r.grad is not None else np.zeros(tuple(r.shape)) for r in ref], axis=1)
        out[f"proj_{which}_radii"] = radii.numpy().astype(np.int32)
        # conditioning of the blurred 2-D covariance, kappa = (a1 + c1)^2 / det1 (>= 4): the fp32 kernel loses
        # ~eps * kappa in det1 = a1 c1 - b^2, and every conic / compensation gradient divides by det1
        con64 = con.detach().numpy()
        det_inv = con64[:, 0] * con64[:, 2] - con64[:, 1] ** 2          # = 1 / det1
        with np.errstate(all="ignore"):
            out[f"proj_{which}_kappa"] = np.where(radii.numpy() > 0, (con64[:, 0] + con64[:, 2]) ** 2 / det_inv, 0.0)
        # rows whose Jacobian is evaluated at the clamp limit
        x = (means.double() @ V[:3, :3].T + V[:3, 3])
        tx, ty = (x[:, 0] / x[:, 2]).abs().numpy(), (x[:, 1] / x[:, 2]).abs().numpy()
        out[f"proj_{which}_is_clamped"] = (tx > 1.3 * 0.5 * 320 / 350.0) | (ty > 1.3 * 0.5 * 200 / 350.0)
    # gradients are stored as float32 (the float64 values rounded once): the drift test compares at 1e-6
    out = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float64 else v) for k, v in out.items()}
    np.savez_compressed(os.path.join(GOLD, "bwd_small.npz"), **out)


def full_size_digest():
    """S-100k at the full 1920x1280 resolution, stored as seed + digests (CRC32 of the
    raw bytes of every integer / bit-exact float tensor, a coarse 8x8-block summary of the image), so the
    GPU test can check a full-resolution frame without running the oracle on the box."""
    import json
    import zlib
    from oracle import gsplat_oracle as O
    from street_crafter_amd.scenes import make_camera, make_scene_portable
    sc = make_scene_portable(100_000)          # bit-identical inputs on every host (see its docstring)
    cam = make_camera()
    crc = lambda a: int(zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff)
    r = O.render_frame(sc.means.numpy(), sc.quats.numpy(), sc.scales.numpy(), sc.opacities.numpy(), sc.sh.numpy(),
                       cam.viewmat.numpy(), cam.K.numpy(), cam.width, cam.height, sc.sh_degree,
                       cam_center=cam.camera_center.numpy(), near_plane=cam.znear, far_plane=cam.zfar,
                       return_unstable=True)
    img = np.concatenate([r["render_colors"][0], r["render_alphas"][0]], axis=-1).astype(np.float64)   # [H,W,5]
    H, W = img.shape[:2]
    # pixels within 2e-5 (relative) of a hard threshold may flip on a 1-ulp exp difference -- numpy's own
    # exp differs by an ulp between AVX2 and AVX-512 hosts -- so they are listed and left out of the means
    stable = ~r["unstable"][0]
    img[~stable] = 0.0
    cnt = stable.reshape(H // 160, 160, W // 240, 240).sum(axis=(1, 3))
    blocks = img.reshape(H // 160, 160, W // 240, 240, 5).sum(axis=(1, 3)) / cnt[..., None]  # [8,8,5]
    d = {"scene": {"n": 100000, "generator": "scenes.make_scene_portable", "width": cam.width, "height": cam.height,
                   "inputs_crc32": {k: crc(getattr(sc, k).numpy()) for k in ("means", "quats", "scales", "opacities", "sh")}},
         "n_isects": int(r["isect_ids"].shape[0]),
         "crc32": {k: crc(r[k]) for k in ("radii", "means2d", "depths", "conics", "compensations", "opacities",
                                          "tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets", "colors")},
         "unstable_yx": np.argwhere(~stable).astype(int).tolist(),
         "image_block_means_over_stable_pixels": np.round(blocks, 8).tolist()}
    json.dump(d, open(os.path.join(GOLD, "s100k_fullres_digest.json"), "w"), indent=1)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    if os.path.isdir(REF):
        ref_vectors()
    else:
        print("reference not present: *_ref.npz left untouched")
    oracle_vectors()
    backward_vectors()
    full_size_digest()
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))
