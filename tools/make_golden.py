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

    # (2)+(3) small full pipeline: 128 x 96 image (8 x 6 tiles), 3000 Gaussians, 4 channels
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
    full_size_digest()
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))
