// Library-level entry points of the C ABI (include/street_crafter_amd.h).
#include "raster_common.h"
#include <string.h>
#include <time.h>

// SC_ABI_HASH: a digest of include/street_crafter_amd.h, handed to BOTH binaries by build.py (this library and the binding
// layer, csrc/binding.cpp): a binding layer compiled against another edition of the header than the library beside it
// (signatures!) refuses to load (_lib.py) instead of calling through stale prototypes.
#ifndef SC_ABI_HASH
#define SC_ABI_HASH "unhashed"
#endif
#ifdef SC_DIAG
extern "C" const char* sc_version(void) { return "street_crafter_amd 0.4.0 (gfx950, DIAGNOSTIC build) abi:" SC_ABI_HASH; }
#else
extern "C" const char* sc_version(void) { return "street_crafter_amd 0.4.0 (gfx950) abi:" SC_ABI_HASH; }
#endif

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

#ifdef SC_DIAG
int g_sc_debug[4] = {0, 0, 0, 0};       // diagnostic build only (sc_common.h)
#endif
int g_sc_proj_clamp = 0;        // sc_set_option "proj_clamp" (projection_common.h)
int g_sc_radius_floor = 0;      // sc_set_option "radius_floor"

// Host-side wait for the sequence number that center_scatter_kernel publishes behind the frame's sizes
// (host-mapped pinned memory, include/street_crafter_amd.h sc_isect_bin_count).  A plain spin in C: a Python host
// calls it through ctypes.CDLL, which releases the GIL for the duration, so a second host thread (the other frame
// in flight) keeps launching while this one waits for its counts.
extern "C" int sc_wait_i64(const int64_t* addr, int64_t value, int64_t timeout_us) {
    if (!addr) return SC_EINVAL;
    struct timespec t0;
    bool have_t0 = false;
    for (unsigned spins = 1;; ++spins) {
        if (__atomic_load_n(addr, __ATOMIC_ACQUIRE) == value) return SC_OK;
        __builtin_ia32_pause();
        if ((spins & 1023u) == 0) {
            struct timespec now;
            clock_gettime(CLOCK_MONOTONIC, &now);
            if (!have_t0) { t0 = now; have_t0 = true; }
            const int64_t us = (int64_t)(now.tv_sec - t0.tv_sec) * 1000000 + (now.tv_nsec - t0.tv_nsec) / 1000;
            if (timeout_us >= 0 && us > timeout_us) return 1;       // timed out (not an error code of the library)
        }
    }
}

extern "C" int sc_stream_create(int priority, const uint32_t* cu_mask, int n_mask_words, sc_stream_t* out) {
    if (!out || n_mask_words < 0 || (n_mask_words > 0 && !cu_mask)) return SC_EINVAL;
    hipStream_t s = nullptr;
    if (n_mask_words > 0) {
        bool any = false;
        for (int i = 0; i < n_mask_words; ++i) any = any || cu_mask[i] != 0u;
        if (!any) return SC_EINVAL;                      // an empty mask would hang the first launch
        SC_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_mask_words, cu_mask));
    } else {
        SC_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority));
    }
    *out = (sc_stream_t)s;
    return SC_OK;
}

extern "C" int sc_stream_destroy(sc_stream_t stream) {
    if (!stream) return SC_EINVAL;
    SC_HIP(hipStreamDestroy(sc_s(stream)));
    return SC_OK;
}

extern "C" int sc_stream_priority_range(int* least, int* greatest) {
    if (!least || !greatest) return SC_EINVAL;
    SC_HIP(hipDeviceGetStreamPriorityRange(least, greatest));
    return SC_OK;
}

extern "C" int sc_set_option(const char* key, int value) {
    if (!key) return SC_EINVAL;
#ifdef SC_DIAG     // "debug0".."debug3" exist in the diagnostic build only: the shipped library answers SC_EINVAL
    if (strncmp(key, "debug", 5) == 0 && key[5] >= '0' && key[5] <= '3' && key[6] == 0) {
        if (value < 0) return SC_EINVAL;
        const int prev = g_sc_debug[key[5] - '0'];
        g_sc_debug[key[5] - '0'] = value;
        return prev;
    }
#endif
    if (strcmp(key, "raster_bwd") == 0) {
        if (value < 0 || value > 1) return SC_EINVAL;
        const int prev = g_sc_raster_bwd_variant;
        g_sc_raster_bwd_variant = value;
        return prev;
    }
    if (strcmp(key, "raster_bwd_split") == 0) {
        if (value < 0 || value > 1) return SC_EINVAL;
        const int prev = g_sc_raster_bwd_split;
        g_sc_raster_bwd_split = value;
        return prev;
    }
    if (strcmp(key, "raster_map") == 0) {
        if (value < 0 || value > 1) return SC_EINVAL;
        const int prev = g_sc_raster_map;
        g_sc_raster_map = value;
        return prev;
    }
    if (strcmp(key, "raster_hint_blend") == 0) {
        if (value < 0 || value > 4) return SC_EINVAL;
        const int prev = g_sc_raster_hint_blend;
        g_sc_raster_hint_blend = value;
        return prev;
    }
    if (strcmp(key, "raster_split") == 0) {
        if (value < 0 || value > 100) return SC_EINVAL;
        const int prev = g_sc_raster_split;
        g_sc_raster_split = value;
        return prev;
    }
    if (strcmp(key, "proj_clamp") == 0) {
        if (value < 0 || value > 1) return SC_EINVAL;
        const int prev = g_sc_proj_clamp;
        g_sc_proj_clamp = value;
        return prev;
    }
    if (strcmp(key, "radius_floor") == 0) {
        if (value < 0 || value > 1) return SC_EINVAL;
        const int prev = g_sc_radius_floor;
        g_sc_radius_floor = value;
        return prev;
    }
    if (strcmp(key, "isect_pull") == 0) {
        if (value < 0 || value > 1) return SC_EINVAL;
        const int prev = g_sc_isect_pull;
        g_sc_isect_pull = value;
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
// The tail of render_novel_view (street_gaussian_renderer.py:151-163) plus the visualizer's uint8
// conversion, as ONE pass over the frame:
//     rgb = clamp(clamp(fg, 0, 1) + clamp(sky, 0, 1) * (1 - acc), 0, 1)     (sky / acc optional)
//     u8  = (uint8)(rgb * 255)          rounding 0: the video frames, (rgb * 255).astype(np.uint8)
//                                       (street_gaussian_visualizer.py:97, base_visualizer.py:37)
//     u8  = (uint8)(rgb * 255 + 0.5)    rounding 1: torchvision.utils.save_image's PNGs
// fg / sky are the rasterizer's raw [H,W,C>=3] images (the per-pass clamp of renderer.py:290-291 is
// folded in).  Every product / sum is a separately rounded fp32 op, so the result is bit-identical to
// the torch composition the reference spells out.
namespace {
__global__ __launch_bounds__(256) void frame_composite_u8_kernel(
    const float* __restrict__ fg, int64_t fg_stride, int64_t fg_ch, const float* __restrict__ acc,
    const float* __restrict__ sky, int64_t sky_stride, int64_t sky_ch, int64_t n_pix, int rounding,
    uint8_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pix) return;
    const float* p = fg + i * fg_stride;
    const float* q = sky ? sky + i * sky_stride : nullptr;
    const float keep = sky ? __fsub_rn(1.0f, acc[i]) : 0.0f;
    const float bias = rounding ? 0.5f : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = fminf(fmaxf(p[c * fg_ch], 0.0f), 1.0f);
        if (sky) {
            const float s = fminf(fmaxf(q[c * sky_ch], 0.0f), 1.0f);
            v = fminf(fmaxf(__fadd_rn(v, __fmul_rn(s, keep)), 0.0f), 1.0f);
        }
        dst[i * 3 + c] = (uint8_t)__fadd_rn(__fmul_rn(v, 255.0f), bias);
    }
}
}  // namespace

extern "C" int sc_frame_composite_u8(const float* fg, int fg_stride, const float* acc, const float* sky,
                                     int sky_stride, int64_t n_pixels, int rounding, uint8_t* out,
                                     sc_stream_t stream) {
    if (n_pixels < 0 || fg_stride < 3 || (sky && sky_stride < 3) || (rounding != 0 && rounding != 1)) return SC_EINVAL;
    if ((sky != nullptr) != (acc != nullptr)) return SC_EINVAL;
    if (n_pixels == 0) return SC_OK;
    if (!fg || !out) return SC_EINVAL;
    hipLaunchKernelGGL(frame_composite_u8_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0,
                       sc_s(stream), fg, (int64_t)fg_stride, (int64_t)1, acc, sky, (int64_t)sky_stride, (int64_t)1, n_pixels,
                       rounding, out);
    SC_LAUNCH_CHECK();
    return SC_OK;
}

// The same with a channel stride: images stored as planes (sc_rasterize_fwd_planar: pixel stride 1, channel stride H * W).
extern "C" int sc_frame_composite_u8_strided(const float* fg, int64_t fg_pix_stride, int64_t fg_ch_stride, const float* acc,
                                             const float* sky, int64_t sky_pix_stride, int64_t sky_ch_stride,
                                             int64_t n_pixels, int rounding, uint8_t* out, sc_stream_t stream) {
    if (n_pixels < 0 || fg_pix_stride < 1 || fg_ch_stride < 1 || (rounding != 0 && rounding != 1)) return SC_EINVAL;
    if (sky && (sky_pix_stride < 1 || sky_ch_stride < 1)) return SC_EINVAL;
    if ((sky != nullptr) != (acc != nullptr)) return SC_EINVAL;
    if (n_pixels == 0) return SC_OK;
    if (!fg || !out) return SC_EINVAL;
    hipLaunchKernelGGL(frame_composite_u8_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0,
                       sc_s(stream), fg, fg_pix_stride, fg_ch_stride, acc, sky, sky_pix_stride, sky_ch_stride, n_pixels,
                       rounding, out);
    SC_LAUNCH_CHECK();
    return SC_OK;
}
