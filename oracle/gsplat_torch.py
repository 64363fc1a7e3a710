"""Differentiable torch-CPU restatement of the gsplat forward ops -- TEST INFRASTRUCTURE ONLY.

Gradient oracle: the HIP backward kernels (SURVEY.md A.1 bwd, A.4 bwd, A.6) are
checked against ``torch.autograd`` run through these forward restatements, i.e.
against an independent derivation of the same VJPs.  The forward here follows
``oracle/gsplat_oracle.py`` (same formulas, dtype selectable so tests can run it in
float64 for a tight reference); hard decisions (cull, alpha skip, termination) are
treated as constants, exactly as the upstream kernels do.

Parity status: see the header of ``oracle/gsplat_oracle.py`` ("parity unpinned"
except the SH basis).  Only tests/, smoke() and bench.py's cpu_baseline may import this.
"""
from __future__ import annotations

import math
import torch

ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.999
T_EPS = 1e-4


def quat_scale_to_covar(quats, scales):
    q = quats / quats.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)
    M = R * scales[:, None, :]
    return M @ M.transpose(1, 2)


def fully_fused_projection(means, quats, scales, viewmat, K, width, height, eps2d=0.3,
                           near_plane=0.01, far_plane=1e10, radius_clip=0.0, proj_clamp="symmetric", radius_floor=0.01):
    """Returns radii (long), means2d, depths, conics, compensations; culled rows zero.
    proj_clamp / radius_floor: the upstream-version-dependent constants (gsplat_oracle.py, PROJ_CLAMPS / RADIUS_FLOORS)."""
    assert proj_clamp in ("symmetric", "asymmetric") and float(radius_floor) in (0.01, 0.1), (proj_clamp, radius_floor)
    Wm = viewmat[:3, :3]
    t = viewmat[:3, 3]
    pc = means @ Wm.T + t
    x, y, z = pc.unbind(-1)
    valid = ~((z < near_plane) | (z > far_plane))
    cov = quat_scale_to_covar(quats, scales)
    covc = Wm @ cov @ Wm.T
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    tanx, tany = 0.5 * width / fx, 0.5 * height / fy
    if proj_clamp == "symmetric":
        limxp = limxn = 1.3 * tanx
        limyp = limyn = 1.3 * tany
    else:
        limxp, limxn = (width - cx) / fx + 0.3 * tanx, cx / fx + 0.3 * tanx
        limyp, limyn = (height - cy) / fy + 0.3 * tany, cy / fy + 0.3 * tany
    zs = torch.where(valid, z, torch.ones_like(z))
    rz = 1.0 / zs
    tx = zs * torch.minimum(torch.maximum(x * rz, -limxn), limxp)
    ty = zs * torch.minimum(torch.maximum(y * rz, -limyn), limyp)
    zero = torch.zeros_like(z)
    J = torch.stack([fx * rz, zero, -fx * tx * rz * rz,
                     zero, fy * rz, -fy * ty * rz * rz], dim=-1).reshape(-1, 2, 3)
    cov2 = J @ covc @ J.transpose(1, 2)
    m2x = fx * x * rz + cx
    m2y = fy * y * rz + cy
    a, b, c = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 1]
    det0 = a * c - b * b
    a1, c1 = a + eps2d, c + eps2d
    det1 = a1 * c1 - b * b
    valid = valid & ~(det1 <= 0)
    det1s = torch.where(valid, det1, torch.ones_like(det1))
    comp = torch.sqrt(torch.clamp(det0 / det1s, min=0.0))
    conics = torch.stack([c1 / det1s, -b / det1s, a1 / det1s], dim=-1)
    bb = 0.5 * (a1 + c1)
    lam = bb + torch.sqrt(torch.clamp(bb * bb - det1s, min=float(radius_floor)))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    valid = valid & ~(radius <= radius_clip)
    valid = valid & ~((m2x + radius <= 0) | (m2x - radius >= width) |
                      (m2y + radius <= 0) | (m2y - radius >= height))
    radii = torch.where(valid, radius, torch.zeros_like(radius)).long()
    vm = valid.to(means.dtype)
    means2d = torch.stack([m2x, m2y], dim=-1) * vm[:, None]
    return radii, means2d, z * vm, conics * vm[:, None], comp * vm


def sh_basis(degree, dirs):
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    x, y, z = d.unbind(-1)
    Y = [torch.full_like(x, 0.2820947917738781)]
    if degree >= 1:
        c1 = 0.48860251190292
        Y += [-c1 * y, c1 * z, -c1 * x]
    if degree >= 2:
        z2 = z * z
        fTmp0B = -1.092548430592079 * z
        fC1 = x * x - y * y
        fS1 = 2.0 * x * y
        pSH6 = 0.9461746957575601 * z2 - 0.3153915652525201
        Y += [0.5462742152960395 * fS1, fTmp0B * y, pSH6, fTmp0B * x, 0.5462742152960395 * fC1]
    if degree >= 3:
        fTmp0C = -2.285228997322329 * z2 + 0.4570457994644658
        fTmp1B = 1.445305721320277 * z
        fC2 = x * fC1 - y * fS1
        fS2 = x * fS1 + y * fC1
        pSH12 = z * (1.865881662950577 * z2 - 1.119528997770346)
        Y += [-0.5900435899266435 * fS2, fTmp1B * fS1, fTmp0C * y, pSH12, fTmp0C * x,
              fTmp1B * fC1, -0.5900435899266435 * fC2]
    if degree >= 4:
        fTmp0D = z * (-4.683325804901025 * z2 + 2.007139630671868)
        fTmp1C = 3.31161143515146 * z2 - 0.47308734787878
        fTmp2B = -1.770130769779931 * z
        fC3 = x * fC2 - y * fS2
        fS3 = x * fS2 + y * fC2
        pSH20 = 1.984313483298443 * z * pSH12 + -1.006230589874905 * pSH6
        Y += [0.6258357354491763 * fS3, fTmp2B * fS2, fTmp1C * fS1, fTmp0D * y, pSH20,
              fTmp0D * x, fTmp1C * fC1, fTmp2B * fC2, 0.6258357354491763 * fC3]
    return torch.stack(Y, dim=-1)


def spherical_harmonics(degree, dirs, coeffs, masks=None):
    Kn = (degree + 1) ** 2
    Y = sh_basis(degree, dirs)
    out = (Y[..., :, None] * coeffs[..., :Kn, :]).sum(dim=-2)
    if masks is not None:
        out = out * masks[..., None].to(out.dtype)
    return out


def rasterize_to_pixels(means2d, conics, colors, opacities, image_width, image_height,
                        tile_size, isect_offsets, flatten_ids, backgrounds=None, pixel_grads=None):
    """Shapes as gsplat: [C,N,*]; returns render_colors [C,H,W,D], render_alphas [C,H,W,1].

    `pixel_grads`: optional list; when given, every tile appends (flat ids, dx, dy) with dx/dy
    [G,P] retaining their gradients, so that after backward `absgrad_from_pixel_grads` can form
    gsplat's absgrad = sum over pixels of |d loss / d mean2d through that pixel|."""
    C, N = opacities.shape
    D = colors.shape[-1]
    H, W = int(image_height), int(image_width)
    offs = isect_offsets.reshape(-1).tolist()
    th, tw = isect_offsets.shape[1], isect_offsets.shape[2]
    I = flatten_ids.shape[0]
    dt = means2d.dtype
    m2 = means2d.reshape(-1, 2)
    cn = conics.reshape(-1, 3)
    co = colors.reshape(-1, D)
    op = opacities.reshape(-1)
    rows_c = []
    rows_a = []
    for c in range(C):
        img_c = torch.zeros(H, W, D, dtype=dt)
        img_a = torch.zeros(H, W, 1, dtype=dt)
        pieces = []
        for ty in range(th):
            for tx in range(tw):
                tflat = (c * th + ty) * tw + tx
                start = offs[tflat]
                end = offs[tflat + 1] if tflat + 1 < len(offs) else I
                y0, x0 = ty * tile_size, tx * tile_size
                y1, x1 = min(y0 + tile_size, H), min(x0 + tile_size, W)
                if y1 <= y0 or x1 <= x0:
                    continue
                hh, ww = y1 - y0, x1 - x0
                P = hh * ww
                if end <= start:
                    if backgrounds is not None:
                        pieces.append((y0, y1, x0, x1,
                                       backgrounds[c][None, :].expand(P, D).reshape(hh, ww, D),
                                       torch.zeros(hh, ww, 1, dtype=dt)))
                    continue
                px = (torch.arange(x0, x1, dtype=dt) + 0.5)[None, :].expand(hh, ww).reshape(-1)
                py = (torch.arange(y0, y1, dtype=dt) + 0.5)[:, None].expand(hh, ww).reshape(-1)
                g = flatten_ids[start:end].long()
                G = g.shape[0]
                dx = m2[g, 0][:, None] - px[None, :]
                dy = m2[g, 1][:, None] - py[None, :]
                if pixel_grads is not None:
                    dx.retain_grad()
                    dy.retain_grad()
                    pixel_grads.append((g, dx, dy))
                ca, cb, cc = cn[g, 0][:, None], cn[g, 1][:, None], cn[g, 2][:, None]
                sigma = 0.5 * (ca * dx * dx + cc * dy * dy) + cb * dx * dy
                alpha_raw = op[g][:, None] * torch.exp(-sigma)
                alpha = torch.clamp(alpha_raw, max=ALPHA_MAX)
                ok = (~((sigma < 0) | (alpha < ALPHA_MIN))).detach()
                a_eff = torch.where(ok, alpha, torch.zeros_like(alpha))
                om = 1.0 - a_eff
                T_after = torch.cumprod(om, dim=0)
                T_before = torch.cat([torch.ones(1, P, dtype=dt), T_after[:-1]], dim=0)
                term = ok & (T_after.detach() <= T_EPS)
                any_term = term.any(dim=0)
                first = torch.where(any_term, term.to(torch.int64).argmax(dim=0),
                                    torch.full((P,), G, dtype=torch.int64))
                kk = torch.arange(G)[:, None]
                live = ok & (kk < first[None, :])
                vis = torch.where(live, a_eff * T_before, torch.zeros_like(a_eff))
                acc = (co[g][:, None, :] * vis[:, :, None]).sum(dim=0)
                om_live = torch.where(live, om, torch.ones_like(om))
                Tfin = torch.prod(om_live, dim=0)
                if backgrounds is not None:
                    acc = acc + Tfin[:, None] * backgrounds[c][None, :]
                pieces.append((y0, y1, x0, x1, acc.reshape(hh, ww, D),
                               (1.0 - Tfin).reshape(hh, ww, 1)))
        # assemble without in-place writes on leaf-dependent tensors
        if pieces:
            pad_c = []
            pad_a = []
            for (y0, y1, x0, x1, pc, pa) in pieces:
                pad_c.append(torch.nn.functional.pad(pc.permute(2, 0, 1),
                                                     (x0, W - x1, y0, H - y1)).permute(1, 2, 0))
                pad_a.append(torch.nn.functional.pad(pa.permute(2, 0, 1),
                                                     (x0, W - x1, y0, H - y1)).permute(1, 2, 0))
            img_c = img_c + torch.stack(pad_c).sum(0)
            img_a = img_a + torch.stack(pad_a).sum(0)
        rows_c.append(img_c)
        rows_a.append(img_a)
    return torch.stack(rows_c), torch.stack(rows_a)


def absgrad_from_pixel_grads(pixel_grads, n_flat):
    """[n_flat, 2] = sum over (tile, pixel) of |grad of dx|, |grad of dy| per flat Gaussian id
    (d dx / d mean_x = 1, so the per-pixel gradient of the mean IS the gradient of dx)."""
    out = torch.zeros(n_flat, 2, dtype=torch.float64)
    for g, dx, dy in pixel_grads:
        if dx.grad is None:
            continue
        out[:, 0].index_add_(0, g, dx.grad.abs().sum(dim=1).double())
        out[:, 1].index_add_(0, g, dy.grad.abs().sum(dim=1).double())
    return out
