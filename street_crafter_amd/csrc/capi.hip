// Library-level entry points of the C ABI (include/street_crafter_amd.h).
#include "raster_common.h"
#include <string.h>

extern "C" const char* sc_version(void) { return "street_crafter_amd 0.1.0 (gfx950)"; }

extern "C" const char* sc_target_arch(void) { return "gfx950"; }

extern "C" const char* sc_error_string(int code) {
    switch (code) {
        case SC_OK: return "ok";
        case SC_EINVAL: return "street_crafter_amd: invalid argument (size / null pointer / unsupported parameter)";
        case SC_EWORKSPACE: return "street_crafter_amd: workspace too small";
        case SC_EUNSUPPORTED: return "street_crafter_amd: unsupported configuration";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "street_crafter_amd: unknown error";
}

int g_sc_debug[4] = {0, 0, 0, 0};

extern "C" int sc_set_option(const char* key, int value) {
    if (!key) return SC_EINVAL;
    if (strncmp(key, "debug", 5) == 0 && key[5] >= '0' && key[5] <= '3' && key[6] == 0) {
        if (value < 0) return SC_EINVAL;
        const int prev = g_sc_debug[key[5] - '0'];
        g_sc_debug[key[5] - '0'] = value;
        return prev;
    }
    if (strcmp(key, "raster_bwd") == 0) {
        if (value < 0 || value > 1) return SC_EINVAL;
        const int prev = g_sc_raster_bwd_variant;
        g_sc_raster_bwd_variant = value;
        return prev;
    }
    if (strcmp(key, "raster_fwd") == 0) {
        if (value != 0 && value != 3) return SC_EINVAL;
        const int prev = g_sc_raster_fwd_variant;
        g_sc_raster_fwd_variant = value;
        return prev;
    }
    return SC_EINVAL;
}

// ---- frame export used by the multi-GPU gather (street_crafter_amd/dist.py) ---------------------
// rgb f32 [C?,H,W,>=3] with channel stride `cstride` floats per pixel -> uint8 [H,W,3]:
// clamp to [0,1], *255, round half up.  One pass instead of five torch elementwise kernels.
namespace {
__global__ __launch_bounds__(256) void frame_to_u8_kernel(const float* __restrict__ src, int64_t n_pix,
                                                          int cstride, uint8_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pix) return;
    const float* p = src + i * cstride;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = fminf(fmaxf(p[c], 0.0f), 1.0f) * 255.0f + 0.5f;
        dst[i * 3 + c] = (uint8_t)v;
    }
}
}  // namespace

extern "C" int sc_frame_to_u8(const float* rgb, int64_t n_pixels, int channel_stride, uint8_t* out,
                              sc_stream_t stream) {
    if (n_pixels < 0 || channel_stride < 3) return SC_EINVAL;
    if (n_pixels == 0) return SC_OK;
    if (!rgb || !out) return SC_EINVAL;
    hipLaunchKernelGGL(frame_to_u8_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, sc_s(stream),
                       rgb, n_pixels, channel_stride, out);
    SC_LAUNCH_CHECK();
    return SC_OK;
}
