// Constants and the pinned pair arithmetic shared by the rasterizer kernels.
#pragma once
#include "sc_common.h"

#define SC_ALPHA_MIN (1.0f / 255.0f)
#define SC_ALPHA_MAX 0.999f
#define SC_T_EPS 1e-4f
#define SC_MAX_CDIM 32
#define SC_LOG2E 1.4426950408889634f
#define SC_HALF_LOG2E 0.7213475204444817f

// The pair evaluation.  The blend loop is VALU-bound (rocprof: the VALU is busy ~80 % of the
// kernel), so the per-pixel work is folded as far as the algebra allows:
//     alpha = min(0.999, op * exp(-sigma)),  sigma = 0.5 (a dx^2 + c dy^2) + b dx dy
//           = min(0.999, exp2(log2(op) - sigma2)),
//     sigma2 = sigma * log2(e) = (A2 dx + B2 dy) dx + (C2 dy) dy,   A2 = a log2(e)/2, B2 = b log2(e),
//                                                                 C2 = c log2(e)/2
// A2/B2/C2/log2(op) are computed once per staged splat; (B2 dy) and (C2 dy) dy once per lane row.
// Every rounding point is pinned with explicit mul/fma intrinsics, so every forward variant and the
// backward replay compute bit-identical sigma2 / alpha / transmittance and always take the same
// skip / terminate decisions (no compiler-chosen contraction).  Relative to the literal formula
// alpha moves by ~2e-6 (relative); pixels stay within the 1e-4 bar of the parity tests.
// Opacity <= 0 (or NaN) gives alpha = 0 / NaN and the splat is skipped (alpha >= 1/255 is false).
struct ScSplat { float mx, my, A2, B2, C2, lop; };
typedef float sc_f2 __attribute__((ext_vector_type(2)));      // pixel pairs for the packed-fp32 VALU ops

__device__ __forceinline__ ScSplat sc_prescale(float mx, float my, float a, float b, float c, float op) {
    ScSplat s;
    s.mx = mx; s.my = my;
    s.A2 = __fmul_rn(a, SC_HALF_LOG2E);
    s.B2 = __fmul_rn(b, SC_LOG2E);
    s.C2 = __fmul_rn(c, SC_HALF_LOG2E);
    s.lop = __log2f(op);
    return s;
}
// per-lane-row terms, shared by the pixels of one row
__device__ __forceinline__ float sc_row_b(float B2, float dy) { return __fmul_rn(B2, dy); }
__device__ __forceinline__ float sc_row_q(float C2, float dy) { return __fmul_rn(__fmul_rn(C2, dy), dy); }
__device__ __forceinline__ float sc_sigma2(float A2, float bdy, float q, float dx) {
    return __fmaf_rn(__fmaf_rn(A2, dx, bdy), dx, q);
}
__device__ __forceinline__ float sc_alpha2(float lop, float sigma2) {
    return fminf(SC_ALPHA_MAX, __builtin_amdgcn_exp2f(__fsub_rn(lop, sigma2)));
}
__device__ __forceinline__ bool sc_valid(float sigma2, float alpha) {
    return !(sigma2 < 0.f) && (alpha >= SC_ALPHA_MIN);
}
__device__ __forceinline__ float sc_next_T(float T, float alpha) { return __fmaf_rn(-alpha, T, T); }

// flatten_ids come from the caller (or, in diagnostic runs, from skipped passes): an out-of-range id
// must never become an out-of-bounds gather.  Returns -1 for ids outside [0, n_splats).
__device__ __forceinline__ int sc_safe_id(int g, int n_splats) {
    return ((unsigned)g < (unsigned)n_splats) ? g : -1;
}

// isect_offsets come from the caller too: a tile's [start, end) is clamped into [0, n_isects], so
// corrupt offsets (negative, past the end, decreasing) can shorten or empty a tile's list but never
// turn into an out-of-bounds read of flatten_ids.
__device__ __forceinline__ void sc_tile_range(const int32_t* __restrict__ isect_offsets, int tflat,
                                              int total_tiles, int n_isects, int& range_start, int& range_end) {
    const int s = isect_offsets[tflat];
    const int e = (tflat + 1 < total_tiles) ? isect_offsets[tflat + 1] : n_isects;
    range_start = min(max(s, 0), n_isects);
    range_end = min(max(e, range_start), n_isects);
}

// ---- exact tile-level cull, shared by forward and backward ------------------------------------
// Minimum of q(x,y) = 0.5*(A x^2 + C y^2) + B x y  over the rectangle [x0,x1] x [y0,y1]
// (coordinates relative to the splat centre).  q is a convex quadratic when the conic is
// positive definite; returns 0 if the centre is inside.
__device__ __forceinline__ float min_quad_on_rect(float A, float Bc, float Cc, float x0, float x1,
                                                  float y0, float y1) {
    if (x0 <= 0.f && x1 >= 0.f && y0 <= 0.f && y1 >= 0.f) return 0.f;
    float best = 3.0e38f;
    // vertical edges x = xe: minimise over y -> y* = -B xe / C clamped
    const float invC = 1.0f / Cc, invA = 1.0f / A;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float xe = e ? x1 : x0;
        float ys = fminf(fmaxf(-Bc * xe * invC, y0), y1);
        best = fminf(best, 0.5f * (A * xe * xe + Cc * ys * ys) + Bc * xe * ys);
        const float ye = e ? y1 : y0;
        float xs = fminf(fmaxf(-Bc * ye * invA, x0), x1);
        best = fminf(best, 0.5f * (A * xs * xs + Cc * ye * ye) + Bc * xs * ye);
    }
    return best;
}

// True only when NO pixel centre inside the rectangle can pass the blend loop's test
// (sigma >= 0 and op*exp(-sigma) >= 1/255, i.e. sigma <= ln(255 op)).  Conservative: the
// threshold is widened by an absolute margin plus a bound on the fp32 rounding error of both
// evaluations of the quadratic (2e-6 * the largest magnitude its terms reach on the rectangle);
// NaN inputs and non positive-definite conics are never dropped (every comparison is false).
__device__ __forceinline__ bool splat_misses_rect(float A, float Bc, float Cc, float op, float x0,
                                                  float x1, float y0, float y1) {
    const float L = __logf(255.0f * op);
    if (L + 1e-3f + 1e-3f * fabsf(L) < 0.f) return true;   // op*255 < 1: alpha < 1/255 everywhere
    const bool pd = (A > 0.f) && (Cc > 0.f) && (A * Cc - Bc * Bc > 0.f);
    if (!pd) return false;
    const float mx = fmaxf(fabsf(x0), fabsf(x1)), my = fmaxf(fabsf(y0), fabsf(y1));
    const float S = A * mx * mx + Cc * my * my + 2.0f * fabsf(Bc) * mx * my;
    const float tau = L + 1e-3f + 1e-3f * fabsf(L) + 2e-6f * S;
    return min_quad_on_rect(A, Bc, Cc, x0, x1, y0, y1) > tau;
}

extern int g_sc_raster_bwd_variant;  // sc_set_option "raster_bwd"
extern int g_sc_raster_bwd_split;    // sc_set_option "raster_bwd_split"
extern "C" int sc_tile_order_len(int total_tiles);     // raster_fwd.hip
int sc_tile_order_fwd_items(int total_tiles);          // raster_fwd.hip
extern int g_sc_raster_map;          // sc_set_option "raster_map"
extern int g_sc_raster_fwd_variant;  // see include/street_crafter_amd.h (sc_set_option "raster_fwd")
