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


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    if os.path.isdir(REF):
        ref_vectors()
    else:
        print("reference not present: *_ref.npz left untouched")
    oracle_vectors()
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))
