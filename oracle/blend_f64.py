"""Float64 TRUTH of the per-pixel alpha blend -- TEST INFRASTRUCTURE ONLY (tests/ and tests/fuzz/ import it).

Why it exists ("What's weak" 2): on ill-conditioned splats -- sigma = (a dx^2 + c dy^2)/2 + b dx dy is a
sum of terms of magnitude S in the hundreds while sigma itself is ~1 -- two fp32 evaluation orders of the SAME formula
differ by more than the 1e-4 pixel bar.  The fp32 oracle (oracle/gsplat_oracle.py, SURVEY A.5's literal order) and
the HIP kernels (pre-scaled conic + FMA, csrc/raster_common.h) are two such orders.  Which of them is "off" can only
be said against arithmetic that has no such error: this file evaluates SURVEY A.5 (street_gaussian_renderer.py:267-280
is the call site) in float64 FROM THE SAME fp32 INPUTS (means2d, conics, colours, opacities exactly as the operator
receives them; every fp32 value is exactly representable in float64), per pixel, sequentially.

Hard thresholds.  A.5 has three discontinuities: `sigma < 0`, `alpha < 1/255` (skip) and `T (1 - alpha) <= 1e-4`
(terminate).  Where the float64 value of the tested quantity sits closer to its threshold than ANY fp32 evaluation can
resolve, both outcomes are legitimate fp32 results.  The walk therefore reports every such NEAR decision, with a
rigorous first-order bound of the fp32 error of the quantity:
    sigma:  k_round * 2^-24 * S,   S = (|a| dx^2 + |c| dy^2) / 2 + |b dx dy|   (k_round roundings of terms <= S)
    alpha:  relative error = sigma's absolute error + 4 * 2^-24 (opacity product, exp)
    T:      accumulated relative error of its factors (1 - alpha_j)
and `outcomes()` enumerates the float64 blend under every combination of the near decisions (a handful at most).
A fp32 implementation is RIGHT on a pixel iff it is within tolerance of one of these outcomes.
"""
from __future__ import annotations

import numpy as np

ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.999            # the fp32 constant 0.999f differs from this by 1.3e-8 (relative): far below every window
T_EPS = 1e-4
EPS24 = 2.0 ** -24


class Walk:
    """One float64 blend of one pixel under a given set of forced decisions."""
    __slots__ = ("color", "alpha", "last", "near", "n_blended", "max_S", "forced")

    def __init__(self):
        self.color = None        # float64 [D] (background folded in when given)
        self.alpha = 0.0         # 1 - T
        self.last = -1           # position in the tile's list of the last blended splat (-1: none)
        self.near = []           # [(k, kind, natural_decision)] near decisions met that were NOT forced, in order
        self.n_blended = 0
        self.max_S = 0.0         # largest term magnitude of sigma among the splats that passed the skip test
        self.forced = {}


def _prepare(px, py, g, m2, cn, co, op):
    g = np.asarray(g, dtype=np.int64)
    m = np.asarray(m2, dtype=np.float64).reshape(-1, 2)[g]
    c = np.asarray(cn, dtype=np.float64).reshape(-1, 3)[g]
    col = np.asarray(co, dtype=np.float64)
    col = col.reshape(-1, col.shape[-1])[g]
    o = np.asarray(op, dtype=np.float64).reshape(-1)[g]
    dx = m[:, 0] - float(px)
    dy = m[:, 1] - float(py)
    with np.errstate(all="ignore"):
        t_a, t_c, t_b = 0.5 * c[:, 0] * dx * dx, 0.5 * c[:, 2] * dy * dy, c[:, 1] * dx * dy
        sigma = t_a + t_c + t_b
        S = np.abs(t_a) + np.abs(t_c) + np.abs(t_b)
        raw = o * np.exp(-sigma)
        alpha = np.minimum(ALPHA_MAX, raw)
    return sigma, S, raw, alpha, col


def walk(px, py, g, m2, cn, co, op, background=None, force=None, k_round=8.0, prepared=None) -> Walk:
    """SURVEY A.5 for the pixel centre (px, py) over the tile's list `g` (flat ids, front to back), in float64.
    `force`: {(k, kind): bool} with kind in {"valid", "term"} -- the decision to take at list position k instead
    of the float64-natural one.  Near decisions that are not forced are taken naturally and reported in `.near`."""
    force = force or {}
    sigma, S, raw, alpha, col = prepared if prepared is not None else _prepare(px, py, g, m2, cn, co, op)
    w = Walk()
    w.forced = dict(force)
    D = col.shape[1] if col.ndim == 2 else 0
    acc = np.zeros(D, dtype=np.float64)
    T = 1.0
    terr = 0.0                       # bound of T's relative fp32 error so far
    for k in range(sigma.shape[0]):
        s, a = float(sigma[k]), float(alpha[k])
        ea = k_round * EPS24 * float(S[k]) + 4.0 * EPS24          # |d sigma| bound == relative bound of raw alpha
        if not np.isfinite(s) or not np.isfinite(a):
            valid = False            # NaN / inf inputs: skipped by every implementation (comparisons are false)
        else:
            valid = (s >= 0.0) and (a >= ALPHA_MIN)
            # (log form: raw = 0 -- a splat thousands of sigmas away, whatever its S -- is never "near")
            r_k = float(raw[k])
            es = k_round * EPS24 * float(S[k])          # sigma's own error bound (exactly 0 when every term is 0)
            near_valid = (es > 0.0 and abs(s) <= es) or (r_k > 0.0 and abs(np.log(r_k / ALPHA_MIN)) <= ea + 2e-7)
            if (k, "valid") in force:
                valid = bool(force[(k, "valid")])
            elif near_valid:
                w.near.append((k, "valid", valid))
        if not valid:
            continue
        w.max_S = max(w.max_S, float(S[k]))
        Tn = T * (1.0 - a)
        terr_n = terr + a * ea / max(1.0 - a, 1e-3) + 2.0 * EPS24
        term = Tn <= T_EPS
        near_term = abs(Tn / T_EPS - 1.0) <= terr_n + 2e-7
        if (k, "term") in force:
            term = bool(force[(k, "term")])
        elif near_term:
            w.near.append((k, "term", term))
        if term:
            break
        acc += col[k] * (a * T)
        T, terr = Tn, terr_n
        w.last = k
        w.n_blended += 1
    if background is not None:
        acc = acc + T * np.asarray(background, dtype=np.float64)
    w.color, w.alpha = acc, 1.0 - T
    return w


def outcomes(px, py, g, m2, cn, co, op, background=None, k_round=8.0, max_outcomes=64):
    """-> [Walk]: the natural float64 blend first, then the blend under every other combination of the near
    decisions (each distinct decision path once; at most `max_outcomes`)."""
    prepared = _prepare(px, py, g, m2, cn, co, op)
    out = []

    def rec(force):
        if len(out) >= max_outcomes:
            return
        r = walk(px, py, g, m2, cn, co, op, background, force, k_round, prepared)
        out.append(r)
        for i, (k, kind, nat) in enumerate(r.near):
            f2 = dict(force)
            for (kj, kindj, natj) in r.near[:i]:
                f2[(kj, kindj)] = natj
            f2[(k, kind)] = not nat
            rec(f2)

    rec({})
    return out


def tile_list(px_i, py_i, tile_size, tile_width, isect_offsets, flatten_ids, cam=0, tile_height=None):
    """The slice of flatten_ids the pixel (px_i, py_i) of camera `cam` blends (its tile's list)."""
    offs = np.asarray(isect_offsets).reshape(-1)
    fl = np.asarray(flatten_ids).reshape(-1)
    th = tile_height if tile_height is not None else offs.shape[0] // tile_width
    t = (cam * th + py_i // tile_size) * tile_width + px_i // tile_size
    s = int(offs[t])
    e = int(offs[t + 1]) if t + 1 < offs.shape[0] else fl.shape[0]
    return s, fl[s:e]


def judge_pixels(pixels, width, tile_size, isect_offsets, flatten_ids, m2, cn, co, op, candidates, scale=None,
                 background=None, k_round=8.0):
    """For every (x, y) in `pixels`: the distance of each candidate's pixel to the float64 truth.
    candidates: {name: (colors [H,W,D], alphas [H,W] or [H,W,1])} fp32 images of one camera.
    -> list of dicts: x, y, n_outcomes, S (largest term magnitude), near (count of near decisions on the natural
       path), and per candidate `name`: err = max over channels of |candidate - nearest outcome| / scale (alpha
       included, unscaled), err_natural = the same against the natural float64 path alone."""
    tw = (int(width) + tile_size - 1) // tile_size
    res = []
    for (x, y) in pixels:
        x, y = int(x), int(y)
        _, g = tile_list(x, y, tile_size, tw, isect_offsets, flatten_ids)
        outs = outcomes(x + 0.5, y + 0.5, g, m2, cn, co, op, background, k_round)
        row = {"x": x, "y": y, "n_outcomes": len(outs), "S": outs[0].max_S, "near": len(outs[0].near),
               "n_list": int(len(g)), "n_blended": outs[0].n_blended}
        for name, (img_c, img_a) in candidates.items():
            c = np.asarray(img_c[y, x], dtype=np.float64)
            a = float(np.asarray(img_a[y, x]).reshape(-1)[0])
            sc = np.ones_like(c) if scale is None else np.asarray(scale, dtype=np.float64)

            def dist(o):
                return max(float((np.abs(c - o.color) / sc).max()), abs(a - o.alpha))

            d = [dist(o) for o in outs]
            row[name] = {"err": min(d), "err_natural": d[0], "outcome": int(np.argmin(d))}
        res.append(row)
    return res


# ---- numpy emulation of the kernels' PINNED fp32 arithmetic (csrc/raster_common.h) ---------------------------------
# For CPU-side analysis only (tests/test_oracle_cpu.py): the GPU tests judge the real kernel output.  fma(a, b, c) is
# emulated as float32(float64(a) * float64(b) + float64(c)) (the product of two fp32 numbers is exact in float64; the
# double rounding of the sum differs from a true FMA in rare half-way cases); exp2 / log2 are numpy's (the hardware's
# v_exp_f32 / v_log_f32 are 1-ulp approximations).
_F = np.float32


def _fma32(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def blend_pixel_pinned_fp32(px, py, g, m2, cn, co, op):
    """One pixel through raster_common.h's order of operations: pre-scaled conic (A2 = a log2(e)/2, B2 = b log2(e),
    C2 = c log2(e)/2), sigma2 = fma(fma(A2, dx, B2 dy), dx, (C2 dy) dy), alpha = min(0.999, exp2(log2(op) - sigma2)),
    T' = fma(-alpha, T, T), colour += c * (alpha T) by fma.  -> (colour f32[D], alpha f32)."""
    g = np.asarray(g, dtype=np.int64)
    m2 = np.asarray(m2, np.float32).reshape(-1, 2)
    cn = np.asarray(cn, np.float32).reshape(-1, 3)
    co = np.asarray(co, np.float32)
    co = co.reshape(-1, co.shape[-1])
    op = np.asarray(op, np.float32).reshape(-1)
    A2 = cn[g, 0] * _F(0.7213475204444817)
    B2 = cn[g, 1] * _F(1.4426950408889634)
    C2 = cn[g, 2] * _F(0.7213475204444817)
    with np.errstate(all="ignore"):
        lop = np.log2(op[g]).astype(np.float32)
        dx = (m2[g, 0] - _F(px)).astype(np.float32)
        dy = (m2[g, 1] - _F(py)).astype(np.float32)
        bdy = (B2 * dy).astype(np.float32)
        q = ((C2 * dy).astype(np.float32) * dy).astype(np.float32)
        sg = _fma32(_fma32(A2, dx, bdy), dx, q)
        al = np.minimum(_F(0.999), np.exp2((lop - sg).astype(np.float32)).astype(np.float32))
    valid = (~(sg < 0)) & (al >= _F(1.0 / 255.0))
    T = _F(1.0)
    acc = np.zeros(co.shape[-1], np.float32)
    for k in np.nonzero(valid)[0]:
        nT = _fma32(-al[k], T, T)[()]
        if nT <= _F(1e-4):
            break
        acc = _fma32(co[g[k]], np.full_like(acc, _F(al[k] * T)), acc)
        T = nT
    return acc, _F(1.0) - T
