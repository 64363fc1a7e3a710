"""ctypes front of oracle/gsplat_oracle_c.c -- TEST INFRASTRUCTURE ONLY.

The multi-core C restatement of the forward path (same normative op order as gsplat_oracle.py; integer
outputs and projection / SH floats bit-identical to it, pixels equal up to libm's expf).  Checker at the
sizes numpy is too slow for, and bench.py's `cpu_baseline` (kind "port").  Built by `make -C oracle`
(__graft_entry__.build() does it).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product path never does.  PARITY STATUS: as gsplat_oracle.py (unpinned for
projection / intersection / rasterize; SH basis pinned to the reference's sh_utils.eval_sh)."""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgsplat_oracle_c.so")
_lib = None


def load(build_if_missing=True):
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and build_if_missing:
        subprocess.run(["make", "-s", "-C", _HERE], check=True)
    lib = C.CDLL(LIB_PATH)
    P, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float
    lib.sco_num_threads.restype = i32
    lib.sco_set_num_threads.argtypes = [i32]
    lib.sco_projection.argtypes = [P, P, P, P, P, i64, i32, i32, f32, f32, f32, f32, P, P, P, P, P]
    lib.sco_projection.restype = None
    lib.sco_projection_v.argtypes = [P, P, P, P, P, i64, i32, i32, f32, f32, f32, f32, i32, f32, P, P, P, P, P]
    lib.sco_projection_v.restype = None
    lib.sco_isect_count.argtypes = [P, P, i64, i32, i32, i32, P]
    lib.sco_isect_count.restype = i64
    lib.sco_isect_emit_sort.argtypes = [P, P, P, i32, i64, i32, i32, i32, P, i64, i32, P, P, P]
    lib.sco_isect_emit_sort.restype = i32
    lib.sco_spherical_harmonics.argtypes = [i32, P, P, P, i64, i32, P]
    lib.sco_spherical_harmonics.restype = None
    lib.sco_rasterize.argtypes = [P, P, P, P, P, i32, i64, i32, i32, i32, i32, i32, i32, P, P, i64, P, P, P, P, f32]
    lib.sco_rasterize.restype = None
    lib.sco_rasterize_cond.argtypes = lib.sco_rasterize.argtypes + [f32, P]
    lib.sco_rasterize_cond.restype = None
    _lib = lib
    return lib


def num_threads() -> int:
    return int(load().sco_num_threads())


def set_num_threads(n: int):
    load().sco_set_num_threads(int(n))


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return None if a is None else a.ctypes.data


def fully_fused_projection(means, quats, scales, viewmat, K, width, height, eps2d=0.3, near_plane=0.01,
                           far_plane=1e10, radius_clip=0.0, proj_clamp="symmetric", radius_floor=0.01):
    assert proj_clamp in ("symmetric", "asymmetric") and float(radius_floor) in (0.01, 0.1), (proj_clamp, radius_floor)
    lib = load()
    means, quats, scales = _c(means, np.float32), _c(quats, np.float32), _c(scales, np.float32)
    V, K = _c(viewmat, np.float32), _c(K, np.float32)
    N = means.shape[0]
    radii = np.empty(N, np.int32)
    m2, d = np.empty((N, 2), np.float32), np.empty(N, np.float32)
    con, comp = np.empty((N, 3), np.float32), np.empty(N, np.float32)
    lib.sco_projection_v(_p(means), _p(quats), _p(scales), _p(V), _p(K), N, int(width), int(height), float(eps2d),
                         float(near_plane), float(far_plane), float(radius_clip), 0 if proj_clamp == "symmetric" else 1,
                         float(radius_floor), _p(radii), _p(m2), _p(d), _p(con), _p(comp))
    return radii, m2, d, con, comp


def isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height, sort=True, n_cameras=1,
                return_offsets=False):
    """means2d [C,N,2], radii [C,N], depths [C,N] -> (tiles_per_gauss, isect_ids, flatten_ids[, offsets])."""
    lib = load()
    m2, r, d = _c(means2d, np.float32), _c(radii, np.int32), _c(depths, np.float32)
    Cn, N = r.shape
    tpg = np.empty((Cn, N), np.int32)
    total = int(lib.sco_isect_count(_p(m2), _p(r), Cn * N, int(tile_size), int(tile_width), int(tile_height), _p(tpg)))
    ids, fids = np.empty(total, np.int64), np.empty(total, np.int32)
    off = np.empty((Cn, tile_height, tile_width), np.int32) if return_offsets else None
    rc = lib.sco_isect_emit_sort(_p(m2), _p(r), _p(d), Cn, N, int(tile_size), int(tile_width), int(tile_height),
                                 _p(tpg), total, int(bool(sort)), _p(ids), _p(fids), _p(off))
    if rc != 0:
        raise RuntimeError(f"sco_isect_emit_sort failed ({rc})")
    return (tpg, ids, fids, off) if return_offsets else (tpg, ids, fids)


def spherical_harmonics(degree, dirs, coeffs, masks=None):
    lib = load()
    dirs, coeffs = _c(dirs, np.float32), _c(coeffs, np.float32)
    M, Kb = dirs.size // 3, coeffs.shape[-2]
    mk = None if masks is None else _c(np.asarray(masks, dtype=bool).reshape(-1), np.uint8)
    out = np.empty(dirs.shape, np.float32)
    lib.sco_spherical_harmonics(int(degree), _p(dirs), _p(coeffs), _p(mk), M, Kb, _p(out))
    return out


def rasterize_to_pixels(means2d, conics, colors, opacities, image_width, image_height, tile_size, isect_offsets,
                        flatten_ids, backgrounds=None, return_unstable=False, unstable_rel=2e-5, unstable_cond=0.0,
                        return_cond_bound=False):
    """-> (render_colors [C,H,W,D], render_alphas [C,H,W,1], last_ids [C,H,W][, unstable bool[C,H,W]]).
    unstable_cond: see gsplat_oracle.rasterize_to_pixels.  return_unstable="codes": the flags as u8 codes
    (bit 0 = unstable under the fixed windows alone, bit 1 = unstable at all) instead of bool.
    return_cond_bound (with return_unstable and unstable_cond > 0): appends f32[C,H,W], the first-order bound of the
    blend's own rounding error per pixel, in units of the colour scale."""
    lib = load()
    m2, con = _c(means2d, np.float32), _c(conics, np.float32)
    col, op = _c(colors, np.float32), _c(opacities, np.float32)
    off, fids = _c(isect_offsets, np.int32), _c(flatten_ids, np.int32)
    Cn, N = op.shape
    D = col.shape[-1]
    assert D <= 32
    th, tw = off.shape[1], off.shape[2]
    H, W = int(image_height), int(image_width)
    bg = None if backgrounds is None else _c(backgrounds, np.float32)
    rc, ra = np.empty((Cn, H, W, D), np.float32), np.empty((Cn, H, W, 1), np.float32)
    last = np.empty((Cn, H, W), np.int32)
    unst = np.zeros((Cn, H, W), np.uint8) if return_unstable else None
    cb = np.zeros((Cn, H, W), np.float32) if (return_cond_bound and return_unstable and unstable_cond > 0) else None
    lib.sco_rasterize_cond(_p(m2), _p(con), _p(col), _p(op), _p(bg), Cn, N, D, W, H, int(tile_size), tw, th, _p(off),
                           _p(fids), fids.shape[0], _p(rc), _p(ra), _p(last), _p(unst), float(unstable_rel),
                           float(unstable_cond), _p(cb))
    if return_unstable:
        u = unst if return_unstable == "codes" else unst.astype(bool)
        return (rc, ra, last, u, cb) if cb is not None else (rc, ra, last, u)
    return rc, ra, last


def render_frame(means, quats, scales, opacities, sh_coeffs, viewmat, K, width, height, sh_degree, cam_center=None,
                 tile_size=16, near_plane=0.001, far_plane=1000.0, eps2d=0.3, antialiasing=True,
                 return_unstable=False, unstable_cond=0.0, return_cond_bound=False, proj_clamp="symmetric", radius_floor=0.01):
    """The caller's sequence (renderer.py:186-302), same outputs as gsplat_oracle.render_frame."""
    radii, m2, d, con, comp = fully_fused_projection(means, quats, scales, viewmat, K, width, height, eps2d=eps2d,
                                                     near_plane=near_plane, far_plane=far_plane, proj_clamp=proj_clamp,
                                                     radius_floor=radius_floor)
    opac = np.asarray(opacities, np.float32).reshape(-1)
    if antialiasing:
        opac = opac * comp
    tw, th = math.ceil(width / float(tile_size)), math.ceil(height / float(tile_size))
    tpg, ids, fids, off = isect_tiles(m2[None], radii[None], d[None], tile_size, tw, th, return_offsets=True)
    if cam_center is None:
        V = np.asarray(viewmat, dtype=np.float64)
        cam_center = (-V[:3, :3].T @ V[:3, 3]).astype(np.float32)
    dirs = np.asarray(means, np.float32) - np.asarray(cam_center, np.float32)[None, :]
    cols = spherical_harmonics(sh_degree, dirs, sh_coeffs, masks=radii > 0)
    cols = np.maximum(cols + np.float32(0.5), np.float32(0.0))
    cols4 = np.concatenate([cols, d[:, None]], axis=-1)
    res = rasterize_to_pixels(m2[None], con[None], cols4[None], opac[None], width, height, tile_size, off, fids,
                              return_unstable=return_unstable, unstable_cond=unstable_cond,
                              return_cond_bound=return_cond_bound)
    rc, ra, last = res[:3]
    extra = {"unstable": res[3]} if return_unstable else {}
    if len(res) > 4:
        extra["cond_bound"] = res[4]
    return dict(**extra, radii=radii, means2d=m2, depths=d, conics=con, compensations=comp, opacities=opac,
                tiles_per_gauss=tpg[0], isect_ids=ids, flatten_ids=fids, isect_offsets=off, colors=cols4,
                render_colors=rc, render_alphas=ra, last_ids=last)
