// Constants shared by the rasterizer kernels.
#pragma once
#include "sc_common.h"

#define SC_ALPHA_MIN (1.0f / 255.0f)
#define SC_ALPHA_MAX 0.999f
#define SC_T_EPS 1e-4f
#define SC_MAX_CDIM 32

// alpha-test threshold in the exponent domain:  o * exp(-sigma) >= 1/255  <=>  sigma <= ln(255 o)
__device__ __forceinline__ float sc_fast_exp(float x) { return __expf(x); }

extern int g_sc_raster_fwd_variant;  // 0 = reference-shaped, 1 = culled (default)
