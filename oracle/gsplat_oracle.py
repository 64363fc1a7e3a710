"""CPU oracle for the gsplat operator path -- TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (numpy, IEEE fp32, no fused multiply-add) of the
algorithms behind the six ``gsplat.rendering`` operators that StreetCrafter calls
from ``street_gaussian/models/street_gaussian_renderer.py:204-280``.  It is the
checker for the HIP kernels in ``street_crafter_amd/csrc``; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``gsplat/``, ``simple_knn/``, ``street_crafter_amd/``) never does.

PARITY STATUS: *parity unpinned* for projection / tile intersection / rasterize.
The arithmetic of these ops lives in the third-party CUDA package ``gsplat``
(``pip install git+https://github.com/dendenxu/gsplat.git``, reference
``README.md:35``; a fork with NO commit pinned) which is absent from
``/root/reference`` and from this image, and the reference holds no tests, golden
vectors or fixtures for the path.  The restatement follows
the published gsplat v1.x algorithm and is anchored on the
reference's own call site (argument meaning, shapes, constants).  The one piece
that IS pinned against reference code is ``spherical_harmonics``: its basis is
checked against ``street_gaussian/utils/sh_utils.py:57-112`` (``eval_sh``),
see ``tools/make_golden.py`` and ``tests/golden/sh_eval_ref.npz``.

Normative op order: every expression below is evaluated in fp32, left to right as
parenthesised, one rounding per operation.  The HIP kernels that feed integer
decisions (radii, tile rectangles, depth sort keys) are compiled with
``-ffp-contract=off`` and written with the same op order, so those outputs are
bit-identical, not merely close.

Version choices where upstream gsplat versions differ:
  U1  Jacobian clamp limit = 1.3 * tan(fov/2)          (v1.0-v1.3 form)
  U2  radius discriminant floor = 0.01
  U3  sort covers bits [0, 32 + tile_bits + cam_bits)
"""
from __future__ import annotations

import math
import numpy as np

F32 = np.float32
ALPHA_MIN = F32(1.0 / 255.0)
ALPHA_MAX = F32(0.999)
T_EPS = F32(1e-4)


def _f(x):
    return np.asarray(x, dtype=np.float32)


def _dot3(a0, b0, a1, b1, a2, b2):
    return (a0 * b0 + a1 * b1) + a2 * b2


# --------------------------------------------------------------------------
# a1  fully_fused_projection   (renderer.py:219-234)
# --------------------------------------------------------------------------
def quat_to_rotmat(quats):
    """wxyz quaternion -> 9 rotation entries; same matrix as the reference's
    ``quaternion_to_matrix`` (street_gaussian/utils/general_utils.py:125-146),
    normalised in-op with 1/sqrt (exact IEEE ops)."""
    q = _f(quats)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    n2 = ((x * x + y * y) + z * z) + w * w
    inv = F32(1.0) / np.sqrt(n2)
    w, x, y, z = w * inv, x * inv, y * inv, z * inv
    x2, y2, z2 = x * x, y * y, z * z
    xy, xz, yz = x * y, x * z, y * z
    wx, wy, wz = w * x, w * y, w * z
    one, two = F32(1.0), F32(2.0)
    R00 = one - two * (y2 + z2)
    R01 = two * (xy - wz)
    R02 = two * (xz + wy)
    R10 = two * (xy + wz)
    R11 = one - two * (x2 + z2)
    R12 = two * (yz - wx)
    R20 = two * (xz - wy)
    R21 = two * (yz + wx)
    R22 = one - two * (x2 + y2)
    return R00, R01, R02, R10, R11, R12, R20, R21, R22


def covar_world(quats, scales):
    """Sigma = (R diag(s)) (R diag(s))^T -> 6 unique entries (general_utils.py:332-341)."""
    R00, R01, R02, R10, R11, R12, R20, R21, R22 = quat_to_rotmat(quats)
    s = _f(scales)
    s0, s1, s2 = s[:, 0], s[:, 1], s[:, 2]
    M00, M01, M02 = R00 * s0, R01 * s1, R02 * s2
    M10, M11, M12 = R10 * s0, R11 * s1, R12 * s2
    M20, M21, M22 = R20 * s0, R21 * s1, R22 * s2
    S00 = _dot3(M00, M00, M01, M01, M02, M02)
    S01 = _dot3(M00, M10, M01, M11, M02, M12)
    S02 = _dot3(M00, M20, M01, M21, M02, M22)
    S11 = _dot3(M10, M10, M11, M11, M12, M12)
    S12 = _dot3(M10, M20, M11, M21, M12, M22)
    S22 = _dot3(M20, M20, M21, M21, M22, M22)
    return S00, S01, S02, S11, S12, S22


# The two constants of SURVEY A.1 that differ between upstream gsplat versions (the reference installs an UNPINNED fork,
# README.md:35): U1 the clamp of x/z, y/z in the EWA Jacobian -- "symmetric" = 1.3 tan(fov/2) on both sides (v1.0-1.3, the
# default here), "asymmetric" = ((W - cx)/fx + 0.3 tan, cx/fx + 0.3 tan) (v1.4+; identical for a centred principal point);
# U2 the floor under the radius discriminant -- 0.01 (v1.x, default) or 0.1 (the original Inria rasterizer / early forks).
# The product selects them with sc_set_option("proj_clamp" / "radius_floor"); every oracle takes the same two arguments.
PROJ_CLAMPS = ("symmetric", "asymmetric")
RADIUS_FLOORS = (0.01, 0.1)


def fully_fused_projection(means, quats, scales, viewmat, K, width, height,
                           eps2d=0.3, near_plane=0.01, far_plane=1e10,
                           radius_clip=0.0, calc_compensations=False, proj_clamp="symmetric", radius_floor=0.01):
    """One camera.  Returns radii i32[N], means2d f32[N,2], depths f32[N],
    conics f32[N,3], compensations f32[N] (always computed; caller drops it when
    calc_compensations is False).  Culled rows are all-zero."""
    assert proj_clamp in PROJ_CLAMPS and float(radius_floor) in RADIUS_FLOORS, (proj_clamp, radius_floor)
    means = _f(means)
    V = _f(viewmat)
    K = _f(K)
    N = means.shape[0]
    with np.errstate(all="ignore"):
        mx, my, mz = means[:, 0], means[:, 1], means[:, 2]
        W00, W01, W02, tx_ = V[0, 0], V[0, 1], V[0, 2], V[0, 3]
        W10, W11, W12, ty_ = V[1, 0], V[1, 1], V[1, 2], V[1, 3]
        W20, W21, W22, tz_ = V[2, 0], V[2, 1], V[2, 2], V[2, 3]
        x = _dot3(W00, mx, W01, my, W02, mz) + tx_
        y = _dot3(W10, mx, W11, my, W12, mz) + ty_
        z = _dot3(W20, mx, W21, my, W22, mz) + tz_
        valid = ~((z < F32(near_plane)) | (z > F32(far_plane)))

        S00, S01, S02, S11, S12, S22 = covar_world(quats, scales)
        # T = W * Sigma
        T00 = _dot3(W00, S00, W01, S01, W02, S02)
        T01 = _dot3(W00, S01, W01, S11, W02, S12)
        T02 = _dot3(W00, S02, W01, S12, W02, S22)
        T10 = _dot3(W10, S00, W11, S01, W12, S02)
        T11 = _dot3(W10, S01, W11, S11, W12, S12)
        T12 = _dot3(W10, S02, W11, S12, W12, S22)
        T20 = _dot3(W20, S00, W21, S01, W22, S02)
        T21 = _dot3(W20, S01, W21, S11, W22, S12)
        T22 = _dot3(W20, S02, W21, S12, W22, S22)
        # Sigma_c = T * W^T (6 unique)
        c00 = _dot3(T00, W00, T01, W01, T02, W02)
        c01 = _dot3(T00, W10, T01, W11, T02, W12)
        c02 = _dot3(T00, W20, T01, W21, T02, W22)
        c11 = _dot3(T10, W10, T11, W11, T12, W12)
        c12 = _dot3(T10, W20, T11, W21, T12, W22)
        c22 = _dot3(T20, W20, T21, W21, T22, W22)

        fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
        tanx = F32(0.5) * F32(width) / fx
        tany = F32(0.5) * F32(height) / fy
        if proj_clamp == "symmetric":
            limxp = limxn = F32(1.3) * tanx
            limyp = limyn = F32(1.3) * tany
        else:
            limxp = (F32(width) - cx) / fx + F32(0.3) * tanx
            limxn = cx / fx + F32(0.3) * tanx
            limyp = (F32(height) - cy) / fy + F32(0.3) * tany
            limyn = cy / fy + F32(0.3) * tany
        rz = F32(1.0) / z
        rz2 = rz * rz
        tx = z * np.minimum(limxp, np.maximum(-limxn, x * rz))
        ty = z * np.minimum(limyp, np.maximum(-limyn, y * rz))
        ja = fx * rz
        jb = ((-fx) * tx) * rz2
        jc = fy * rz
        jd = ((-fy) * ty) * rz2
        u0 = ja * c00 + jb * c02
        u1 = ja * c01 + jb * c12
        u2 = ja * c02 + jb * c22
        v1 = jc * c11 + jd * c12
        v2 = jc * c12 + jd * c22
        a = u0 * ja + u2 * jb
        b = u1 * jc + u2 * jd
        c = v1 * jc + v2 * jd
        m2x = (fx * x) * rz + cx
        m2y = (fy * y) * rz + cy

        det0 = a * c - b * b
        a1 = a + F32(eps2d)
        c1 = c + F32(eps2d)
        det1 = a1 * c1 - b * b
        comp = np.sqrt(np.maximum(F32(0.0), det0 / det1))
        valid &= ~(det1 <= F32(0.0))
        valid &= ~np.isnan(det1)

        con0 = c1 / det1
        con1 = (-b) / det1
        con2 = a1 / det1

        bb = F32(0.5) * (a1 + c1)
        lam = bb + np.sqrt(np.maximum(F32(radius_floor), bb * bb - det1))
        radius = np.ceil(F32(3.0) * np.sqrt(lam))
        valid &= ~(radius <= F32(radius_clip))
        valid &= ~np.isnan(radius)
        Wf, Hf = F32(width), F32(height)
        valid &= ~((m2x + radius <= F32(0.0)) | (m2x - radius >= Wf) |
                   (m2y + radius <= F32(0.0)) | (m2y - radius >= Hf))

        radius_c = np.where(valid, np.minimum(radius, F32(2147483520.0)), F32(0.0))
    radii = radius_c.astype(np.int32)
    zero = F32(0.0)
    means2d = np.stack([np.where(valid, m2x, zero), np.where(valid, m2y, zero)], axis=-1)
    depths = np.where(valid, z, zero)
    conics = np.stack([np.where(valid, con0, zero), np.where(valid, con1, zero),
                       np.where(valid, con2, zero)], axis=-1)
    comps = np.where(valid, comp, zero)
    return (radii, means2d.astype(np.float32), depths.astype(np.float32),
            conics.astype(np.float32), comps.astype(np.float32))


# --------------------------------------------------------------------------
# a3  isect_tiles   (renderer.py:241-252)
# --------------------------------------------------------------------------
def tile_bits(n_tiles: int) -> int:
    return int(math.floor(math.log2(n_tiles))) + 1 if n_tiles > 0 else 1


def tile_rects(means2d, radii, tile_size, tile_width, tile_height):
    m = _f(means2d)
    r = np.asarray(radii)
    ts = F32(tile_size)
    tr = r.astype(np.float32) / ts
    tx = m[..., 0] / ts
    ty = m[..., 1] / ts
    tw, th = F32(tile_width), F32(tile_height)
    zero = F32(0.0)
    # fmax(fmin(v, tiles), 0): NaN -> empty rectangle, same as the kernel's fminf/fmaxf
    x0 = np.fmax(np.fmin(np.floor(tx - tr), tw), zero).astype(np.int64)
    x1 = np.fmax(np.fmin(np.ceil(tx + tr), tw), zero).astype(np.int64)
    y0 = np.fmax(np.fmin(np.floor(ty - tr), th), zero).astype(np.int64)
    y1 = np.fmax(np.fmin(np.ceil(ty + tr), th), zero).astype(np.int64)
    vis = r > 0
    x0, x1, y0, y1 = [np.where(vis, v, 0) for v in (x0, x1, y0, y1)]
    return x0, x1, y0, y1


def isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height,
                sort=True, n_cameras=1):
    """means2d [C,N,2], radii [C,N], depths [C,N] -> (tiles_per_gauss i32[C,N],
    isect_ids i64[I], flatten_ids i32[I])."""
    means2d = _f(means2d)
    radii = np.asarray(radii, dtype=np.int32)
    depths = _f(depths)
    C, N = radii.shape
    x0, x1, y0, y1 = tile_rects(means2d, radii, tile_size, tile_width, tile_height)
    tpg = ((y1 - y0) * (x1 - x0)).astype(np.int32)
    tb = tile_bits(tile_width * tile_height)
    flat_tpg = tpg.reshape(-1).astype(np.int64)
    total = int(flat_tpg.sum())
    isect_ids = np.empty(total, dtype=np.int64)
    flatten_ids = np.empty(total, dtype=np.int32)
    if total:
        starts = np.cumsum(flat_tpg) - flat_tpg
        owner = np.repeat(np.arange(C * N, dtype=np.int64), flat_tpg)
        local = np.arange(total, dtype=np.int64) - starts[owner]
        w = (x1 - x0).reshape(-1)[owner]
        ti = y0.reshape(-1)[owner] + local // np.maximum(w, 1)
        tj = x0.reshape(-1)[owner] + local % np.maximum(w, 1)
        cid = owner // N
        dbits = depths.reshape(-1).view(np.uint32).astype(np.int64)[owner]
        isect_ids[:] = (cid << (32 + tb)) | ((ti * tile_width + tj) << 32) | dbits
        flatten_ids[:] = owner.astype(np.int32)
        if sort:
            order = np.argsort(isect_ids, kind="stable")
            isect_ids = isect_ids[order]
            flatten_ids = flatten_ids[order]
    return tpg, isect_ids, flatten_ids


# --------------------------------------------------------------------------
# a4  isect_offset_encode   (renderer.py:253)
# --------------------------------------------------------------------------
def isect_offset_encode(isect_ids, n_cameras, tile_width, tile_height):
    ids = np.asarray(isect_ids, dtype=np.int64)
    n_tiles = tile_width * tile_height
    tb = tile_bits(n_tiles)
    cid = ids >> (32 + tb)
    tid = (ids >> 32) & ((1 << tb) - 1)
    flat = cid * n_tiles + tid
    q = np.arange(n_cameras * n_tiles, dtype=np.int64)
    off = np.searchsorted(flat, q, side="left").astype(np.int32)
    return off.reshape(n_cameras, tile_height, tile_width)


# --------------------------------------------------------------------------
# a6  spherical_harmonics   (renderer.py:259)
# --------------------------------------------------------------------------
def sh_basis(degree, dirs):
    """Real SH basis values Y_k(dir/|dir|), k < (degree+1)^2, list of f32[N]."""
    d = _f(dirs)
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    Y = [np.full(x.shape, F32(0.2820947917738781), dtype=np.float32)]
    if degree < 1:
        return Y
    with np.errstate(all="ignore"):
        inorm = F32(1.0) / np.sqrt((x * x + y * y) + z * z)
    x, y, z = x * inorm, y * inorm, z * inorm
    c1 = F32(0.48860251190292)
    Y += [(-c1) * y, c1 * z, (-c1) * x]
    if degree < 2:
        return Y
    z2 = z * z
    fTmp0B = F32(-1.092548430592079) * z
    fC1 = x * x - y * y
    fS1 = F32(2.0) * x * y
    pSH6 = F32(0.9461746957575601) * z2 - F32(0.3153915652525201)
    pSH7 = fTmp0B * x
    pSH5 = fTmp0B * y
    pSH8 = F32(0.5462742152960395) * fC1
    pSH4 = F32(0.5462742152960395) * fS1
    Y += [pSH4, pSH5, pSH6, pSH7, pSH8]
    if degree < 3:
        return Y
    fTmp0C = F32(-2.285228997322329) * z2 + F32(0.4570457994644658)
    fTmp1B = F32(1.445305721320277) * z
    fC2 = x * fC1 - y * fS1
    fS2 = x * fS1 + y * fC1
    pSH12 = z * (F32(1.865881662950577) * z2 - F32(1.119528997770346))
    pSH13 = fTmp0C * x
    pSH11 = fTmp0C * y
    pSH14 = fTmp1B * fC1
    pSH10 = fTmp1B * fS1
    pSH15 = F32(-0.5900435899266435) * fC2
    pSH9 = F32(-0.5900435899266435) * fS2
    Y += [pSH9, pSH10, pSH11, pSH12, pSH13, pSH14, pSH15]
    if degree < 4:
        return Y
    fTmp0D = z * (F32(-4.683325804901025) * z2 + F32(2.007139630671868))
    fTmp1C = F32(3.31161143515146) * z2 - F32(0.47308734787878)
    fTmp2B = F32(-1.770130769779931) * z
    fC3 = x * fC2 - y * fS2
    fS3 = x * fS2 + y * fC2
    pSH20 = F32(1.984313483298443) * z * pSH12 + F32(-1.006230589874905) * pSH6
    pSH21 = fTmp0D * x
    pSH19 = fTmp0D * y
    pSH22 = fTmp1C * fC1
    pSH18 = fTmp1C * fS1
    pSH23 = fTmp2B * fC2
    pSH17 = fTmp2B * fS2
    pSH24 = F32(0.6258357354491763) * fC3
    pSH16 = F32(0.6258357354491763) * fS3
    Y += [pSH16, pSH17, pSH18, pSH19, pSH20, pSH21, pSH22, pSH23, pSH24]
    return Y


def spherical_harmonics(degree, dirs, coeffs, masks=None):
    """dirs [...,3], coeffs [...,K,3] -> colors [...,3]; rows with masks False are 0.
    Sum order: k ascending, one multiply + one add per term."""
    coeffs = _f(coeffs)
    Y = sh_basis(degree, dirs)
    out = Y[0][..., None] * coeffs[..., 0, :]
    for k in range(1, (degree + 1) ** 2):
        out = out + Y[k][..., None] * coeffs[..., k, :]
    if masks is not None:
        out = np.where(np.asarray(masks, dtype=bool)[..., None], out, F32(0.0))
    return out.astype(np.float32)


# --------------------------------------------------------------------------
# a9  rasterize_to_pixels (forward)   (renderer.py:267-280)
# --------------------------------------------------------------------------
def rasterize_to_pixels(means2d, conics, colors, opacities, image_width, image_height,
                        tile_size, isect_offsets, flatten_ids, backgrounds=None,
                        masks=None, return_unstable=False, unstable_rel=2e-5, unstable_cond=0.0,
                        return_cond_bound=False):
    """means2d [C,N,2], conics [C,N,3], colors [C,N,D], opacities [C,N],
    isect_offsets i32[C,th,tw], flatten_ids i32[I].
    Returns render_colors f32[C,H,W,D], render_alphas f32[C,H,W,1], last_ids i32[C,H,W]
    (+ unstable bool[C,H,W] when asked: pixels where some alpha or transmittance sits
    within `unstable_rel` of a hard threshold, so that a 1-ulp change of exp() may
    legitimately flip the skip / terminate decision).
    `unstable_cond` > 0 (optional; the fixtures and digests were made with 0) widens each window by the
    rounding-error bound of the quantity it guards: sigma is a sum of terms of magnitude
    S = (|A| dx^2 + |C| dy^2) / 2 + |B dx dy|, so another order of the same operations moves it by about
    unstable_cond * 2^-24 * S (S >> sigma for a big rotated splat), alpha by that relative amount and the
    transmittance by the accumulated relative errors of its factors.
    `return_cond_bound` (with return_unstable and unstable_cond > 0) appends f32[C,H,W]: the first-order bound of
    what the same rounding error does to the blend itself, sum_i vis_i (ea_i + 3 eps + terr_i) + T terr, in units
    of the colour scale (ea_i: bound of alpha_i's relative error, terr_i: of the transmittance up to splat i)."""
    means2d = _f(means2d)
    conics = _f(conics)
    colors = _f(colors)
    opacities = _f(opacities)
    offs = np.asarray(isect_offsets, dtype=np.int64)
    fids = np.asarray(flatten_ids, dtype=np.int64)
    C, N = opacities.shape
    D = colors.shape[-1]
    H, W = int(image_height), int(image_width)
    th, tw = offs.shape[1], offs.shape[2]
    I = fids.shape[0]
    out_c = np.zeros((C, H, W, D), dtype=np.float32)
    out_a = np.zeros((C, H, W, 1), dtype=np.float32)
    out_l = np.zeros((C, H, W), dtype=np.int32)
    unstable = np.zeros((C, H, W), dtype=bool)
    want_bound = bool(return_cond_bound and return_unstable and unstable_cond > 0)
    cond_bound = np.zeros((C, H, W), dtype=np.float32)
    flat_off = offs.reshape(-1)
    m2 = means2d.reshape(-1, 2)
    cn = conics.reshape(-1, 3)
    co = colors.reshape(-1, D)
    op = opacities.reshape(-1)
    for c in range(C):
        for ty in range(th):
            for tx in range(tw):
                tflat = (c * th + ty) * tw + tx
                start = int(flat_off[tflat])
                end = int(flat_off[tflat + 1]) if tflat + 1 < flat_off.shape[0] else I
                y0, x0 = ty * tile_size, tx * tile_size
                y1, x1 = min(y0 + tile_size, H), min(x0 + tile_size, W)
                if y1 <= y0 or x1 <= x0:
                    continue
                hh, ww = y1 - y0, x1 - x0
                if masks is not None and not bool(np.asarray(masks)[c, ty, tx]):
                    if backgrounds is not None:
                        out_c[c, y0:y1, x0:x1, :] = _f(backgrounds)[c]
                    continue
                px = (np.arange(x0, x1, dtype=np.float32) + F32(0.5))[None, :].repeat(hh, 0).reshape(-1)
                py = (np.arange(y0, y1, dtype=np.float32) + F32(0.5))[:, None].repeat(ww, 1).reshape(-1)
                P = px.shape[0]
                g = fids[start:end]
                G = g.shape[0]
                if G == 0:
                    Tfin = np.ones(P, dtype=np.float32)
                    acc = np.zeros((P, D), dtype=np.float32)
                    last = np.zeros(P, dtype=np.int32)
                    unst = np.zeros(P, dtype=bool)
                else:
                    with np.errstate(all="ignore"):
                        dx = m2[g, 0][:, None] - px[None, :]
                        dy = m2[g, 1][:, None] - py[None, :]
                        ca, cb, cc = cn[g, 0][:, None], cn[g, 1][:, None], cn[g, 2][:, None]
                        sigma = F32(0.5) * ((ca * dx) * dx + (cc * dy) * dy) + (cb * dx) * dy
                        alpha = np.minimum(ALPHA_MAX, op[g][:, None] * np.exp(-sigma))
                    ok = ~((sigma < F32(0.0)) | (alpha < ALPHA_MIN)) & ~np.isnan(alpha)
                    a_eff = np.where(ok, alpha, F32(0.0))
                    om = F32(1.0) - a_eff
                    # T_after[k] = prod_{j<=k} (1 - a_j), sequential fp32 products
                    T_after = np.cumprod(om, axis=0, dtype=np.float32)
                    T_before = np.concatenate([np.ones((1, P), np.float32), T_after[:-1]], axis=0)
                    term = ok & (T_after <= T_EPS)
                    any_term = term.any(axis=0)
                    stop = np.where(any_term, term.argmax(axis=0), G)  # first terminating k
                    kk = np.arange(G)[:, None]
                    live = ok & (kk < stop[None, :])
                    vis = np.where(live, a_eff * T_before, F32(0.0))
                    contrib = co[g][:, None, :] * vis[:, :, None]          # [G,P,D]
                    acc = np.cumsum(contrib, axis=0, dtype=np.float32)[-1]
                    idx_last = np.where(live, kk, -1).max(axis=0)
                    last = np.where(idx_last >= 0, start + idx_last, 0).astype(np.int32)
                    # transmittance after the last blended splat
                    Tfin = np.where(stop > 0,
                                    T_after[np.clip(stop - 1, 0, G - 1), np.arange(P)],
                                    F32(1.0)).astype(np.float32)
                    if return_unstable:
                        considered = kk <= np.minimum(stop, G - 1)[None, :]
                        if unstable_cond > 0:
                            with np.errstate(all="ignore"):
                                eps24 = F32(2.0 ** -24)
                                ce = F32(unstable_cond) * eps24
                                S = F32(0.5) * (np.abs((ca * dx) * dx) + np.abs((cc * dy) * dy)) + np.abs((cb * dx) * dy)
                                ea = ce * S          # error bound of sigma == relative error bound of alpha
                                wa = F32(unstable_rel) + ea
                                ws = F32(1e-6) + ea
                                t_terms = np.where(ok, (a_eff * (ea + F32(3.0) * eps24)) / om + eps24, F32(0.0))
                                wt = F32(unstable_rel) * F32(10.0) + np.cumsum(t_terms, axis=0, dtype=np.float32)
                                ra = np.abs(alpha - ALPHA_MIN) <= wa * ALPHA_MIN
                                rs = np.abs(sigma) <= ws
                                rt = ok & (np.abs(T_after - T_EPS) <= wt * T_EPS)
                                if want_bound:
                                    terr = np.cumsum(t_terms, axis=0, dtype=np.float32)
                                    b_terms = np.where(live, vis * ((ea + F32(3.0) * eps24) + terr), F32(0.0))
                                    terr_fin = terr[np.minimum(stop, G - 1), np.arange(P)]
                                    bnd = np.cumsum(b_terms, axis=0, dtype=np.float32)[-1] + Tfin * terr_fin
                        else:
                            ra = np.abs(alpha - ALPHA_MIN) <= F32(unstable_rel) * ALPHA_MIN
                            rs = np.abs(sigma) <= F32(1e-6)
                            rt = ok & (np.abs(T_after - T_EPS) <= F32(unstable_rel * 10) * T_EPS)
                        unst = ((ra | rs | rt) & considered).any(axis=0)
                    else:
                        unst = np.zeros(P, dtype=bool)
                if backgrounds is not None:
                    acc = acc + Tfin[:, None] * _f(backgrounds)[c][None, :]
                out_c[c, y0:y1, x0:x1, :] = acc.reshape(hh, ww, D)
                out_a[c, y0:y1, x0:x1, 0] = (F32(1.0) - Tfin).reshape(hh, ww)
                out_l[c, y0:y1, x0:x1] = last.reshape(hh, ww)
                unstable[c, y0:y1, x0:x1] = unst.reshape(hh, ww)
                if want_bound and G > 0:
                    cond_bound[c, y0:y1, x0:x1] = bnd.reshape(hh, ww)
    if want_bound:
        return out_c, out_a, out_l, unstable, cond_bound
    if return_unstable:
        return out_c, out_a, out_l, unstable
    return out_c, out_a, out_l


# --------------------------------------------------------------------------
# the caller's sequence  (renderer.py:186-302), used by tests / smoke / bench
# --------------------------------------------------------------------------
def render_frame(means, quats, scales, opacities, sh_coeffs, viewmat, K, width, height,
                 sh_degree, cam_center=None, tile_size=16, near_plane=0.001, far_plane=1000.0,
                 eps2d=0.3, antialiasing=True, return_unstable=False, unstable_cond=0.0, return_cond_bound=False,
                 proj_clamp="symmetric", radius_floor=0.01):
    """Restates render_kernel_gsplat's forward (one camera).  opacities f32[N,1] or [N]."""
    radii, means2d, depths, conics, comps = fully_fused_projection(
        means, quats, scales, viewmat, K, width, height, eps2d=eps2d,
        near_plane=near_plane, far_plane=far_plane, calc_compensations=antialiasing, proj_clamp=proj_clamp,
        radius_floor=radius_floor)
    opac = _f(opacities).reshape(-1)
    if antialiasing:
        opac = opac * comps
    tw = math.ceil(width / float(tile_size))
    th = math.ceil(height / float(tile_size))
    tpg, isect_ids, flatten_ids = isect_tiles(means2d[None], radii[None], depths[None],
                                              tile_size, tw, th, n_cameras=1)
    offsets = isect_offset_encode(isect_ids, 1, tw, th)
    if cam_center is None:
        V = np.asarray(viewmat, dtype=np.float64)
        cam_center = (-V[:3, :3].T @ V[:3, 3]).astype(np.float32)
    dirs = _f(means) - _f(cam_center)[None, :]
    cols = spherical_harmonics(sh_degree, dirs, sh_coeffs, masks=radii > 0)
    cols = np.maximum(cols + F32(0.5), F32(0.0))
    cols4 = np.concatenate([cols, depths[:, None]], axis=-1)
    res = rasterize_to_pixels(means2d[None], conics[None], cols4[None], opac[None], width, height,
                              tile_size, offsets, flatten_ids, return_unstable=return_unstable,
                              unstable_cond=unstable_cond, return_cond_bound=return_cond_bound)
    out = dict(radii=radii, means2d=means2d, depths=depths, conics=conics, compensations=comps,
               opacities=opac, tiles_per_gauss=tpg[0], isect_ids=isect_ids,
               flatten_ids=flatten_ids, isect_offsets=offsets, colors=cols4,
               render_colors=res[0], render_alphas=res[1], last_ids=res[2])
    if return_unstable:
        out["unstable"] = res[3]
    if len(res) > 4:
        out["cond_bound"] = res[4]
    return out


def psnr(img1, img2):
    """20 log10(1/sqrt(mse)), the reference's definition (utils/loss_utils.py:63-81, no mask)."""
    mse = float(np.mean((np.asarray(img1, np.float64) - np.asarray(img2, np.float64)) ** 2))
    if mse == 0.0:
        return float("inf")
    return 20.0 * math.log10(1.0 / math.sqrt(mse))


# --------------------------------------------------------------------------
# render_novel_view's two-pass frame  (renderer.py:136-163, render_sky :80-93)
# --------------------------------------------------------------------------
def composite_sky(fg_rgb, fg_acc, sky_rgb):
    """rgb = clamp(clamp(fg) + clamp(sky) * (1 - acc), 0, 1): each pass is clamped by render_kernel_gsplat
    in every mode but train (renderer.py:290-291), then `rgb + rgb_sky * (1 - acc)` (:152) and the final
    clamp (:159).  fp32, one rounding per operation.  fg_rgb / sky_rgb [...,3], fg_acc [...,1] or [...]."""
    one, zero = F32(1.0), F32(0.0)
    fg = np.clip(_f(fg_rgb), zero, one)
    sky = np.clip(_f(sky_rgb), zero, one)
    acc = _f(fg_acc)
    if acc.ndim == fg.ndim - 1:
        acc = acc[..., None]
    return np.clip(fg + sky * (one - acc), zero, one).astype(np.float32)


def quantise_u8(rgb, rounding="video"):
    """[0,1] float image -> uint8 the way the reference's visualizer does it: "video" =
    `(rgb * 255).astype(np.uint8)` (street_gaussian_visualizer.py:97; base_visualizer.py:37),
    "save_image" = torchvision.utils.save_image's `mul(255).add_(0.5).clamp_(0, 255).to(uint8)`."""
    v = _f(rgb) * F32(255.0)
    if rounding == "save_image":
        v = np.clip(v + F32(0.5), F32(0.0), F32(255.0))
    elif rounding != "video":
        raise ValueError(rounding)
    return v.astype(np.uint8)


def render_novel_view(fg, sky, viewmat, K, width, height, **kw):
    """fg / sky: dicts with means, quats, scales, opacities, sh, sh_degree (sky may be None).
    -> dict(rgb [H,W,3] composited + clamped, fg=render_frame(fg), sky=render_frame(sky) | None)."""
    def one(s):
        return render_frame(s["means"], s["quats"], s["scales"], s["opacities"], s["sh"], viewmat, K, width,
                            height, s["sh_degree"], **kw)
    rf = one(fg)
    if sky is None:
        rgb = np.clip(rf["render_colors"][0, ..., :3], F32(0.0), F32(1.0))
        return dict(rgb=rgb, fg=rf, sky=None)
    rs = one(sky)
    rgb = composite_sky(rf["render_colors"][0, ..., :3], rf["render_alphas"][0], rs["render_colors"][0, ..., :3])
    return dict(rgb=rgb, fg=rf, sky=rs)
