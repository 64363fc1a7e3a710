// Constants shared by the rasterizer kernels.
#pragma once
#include "sc_common.h"

#define SC_ALPHA_MIN (1.0f / 255.0f)
#define SC_ALPHA_MAX 0.999f
#define SC_T_EPS 1e-4f
#define SC_MAX_CDIM 32

// alpha-test threshold in the exponent domain:  o * exp(-sigma) >= 1/255  <=>  sigma <= ln(255 o)
__device__ __forceinline__ float sc_fast_exp(float x) { return __expf(x); }

// The pair evaluation, with the rounding points pinned by explicit mul/fma intrinsics: every
// forward variant and the backward replay compute bit-identical sigma / alpha / transmittance,
// so their skip / terminate decisions always agree (no compiler-chosen contraction).
//   sigma = 0.5*(a dx^2 + c dy^2) + b dx dy          
__device__ __forceinline__ float sc_sigma(float ca, float cb, float cc, float dx, float dy) {
    const float q = __fmaf_rn(__fmul_rn(cc, dy), dy, __fmul_rn(__fmul_rn(ca, dx), dx));
    return __fmaf_rn(__fmul_rn(cb, dx), dy, __fmul_rn(0.5f, q));
}
__device__ __forceinline__ float sc_vis(float sigma) { return sc_fast_exp(-sigma); }
__device__ __forceinline__ float sc_alpha(float op, float vis) {
    return fminf(SC_ALPHA_MAX, __fmul_rn(op, vis));
}
__device__ __forceinline__ float sc_next_T(float T, float alpha) {
    return __fmul_rn(T, __fsub_rn(1.0f, alpha));
}

extern int g_sc_raster_fwd_variant;  // 0 = reference-shaped, 1 = culled (default)
