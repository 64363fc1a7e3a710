// Shared host/device helpers for the gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/street_crafter_amd.h"

#define SC_WAVE 64

#define SC_LAUNCH_CHECK()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return (int)e__;             \
    } while (0)

#define SC_HIP(call)                                        \
    do {                                                    \
        hipError_t e__ = (call);                            \
        if (e__ != hipSuccess) return (int)e__;             \
    } while (0)

// DIAGNOSTIC BUILD ONLY (-DSC_DIAG: `python -m street_crafter_amd.build --diag` ->
// lib/libstreet_crafter_hip_diag.so, loaded explicitly by tools/exp_*.py through _lib.use_diagnostic_build()).
// The shipped library carries NO diagnostic switch: without SC_DIAG the macros below remove the `dbg` kernel
// parameters, the host-side knob array and every skip branch from the compiled code, and
// sc_set_option("debug0".."debug3") is an unknown key.  (Both GPU memory faults of rounds 1 and 2 were these
// knobs reaching production launches: DESIGN.md section 8.)
// In the diagnostic build a non-zero knob makes a kernel SKIP part of its work so that the part can be priced;
// outputs are then invalid.  RULE: a skip must leave every index that is derived from the skipped producer IN
// BOUNDS for every consumer -- skip stores, never the computation of counts / offsets other code indexes with,
// and bound-check at the consumer anyway.
#ifdef SC_DIAG
extern int g_sc_debug[4];
#define SC_DIAG_PARAM(name) , int name            /* trailing kernel / function parameter */
#define SC_DIAG_ARG(value) , (value)              /* ... and its argument at the call / launch site */
#define SC_DIAG_BIT(name, mask) (((name) & (mask)) != 0)
/* shader clock for phase timing: raw s_memtime (the builtin makes the compiler drain every outstanding load first, which
   serialises exactly the prefetches whose overlap is being measured); SC_DIAG_DRAIN waits for the loads explicitly */
__device__ __forceinline__ unsigned long long sc_diag_clock_raw() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
#define SC_DIAG_CLOCK(on) ((on) ? sc_diag_clock_raw() : 0ull)
__device__ __forceinline__ unsigned long long sc_diag_realtime_raw() {      /* constant 100 MHz */
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
#define SC_DIAG_REALTIME(on) ((on) ? sc_diag_realtime_raw() : 0ull)
#define SC_DIAG_DRAIN(on) do { if (on) asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); } while (0)
#else
#define SC_DIAG_PARAM(name)
#define SC_DIAG_ARG(value)
#define SC_DIAG_BIT(name, mask) false
#define SC_DIAG_CLOCK(on) 0ull
#define SC_DIAG_REALTIME(on) 0ull
#define SC_DIAG_DRAIN(on) do { } while (0)
#endif
extern "C" int sc_tile_order_len(int total_tiles);      // raster_fwd.hip
// VIEW SLOTS: the rasterizer's work hint is kept per VIEW (a street rig renders front / front-left / front-right in
// turn: a frame must not find the hint another camera left).  The count launch (view_slot_lookup, isect_bin.hip)
// matches camera 0's forward axis against a small device-side registry and gives the call a slot number; the hint
// buffer is SC_VIEW_SLOTS banks of C * T words, and the slot travels to the rasterizer in the last word of the
// dispatch list.
constexpr int SC_VIEW_SLOTS = 8;
constexpr int SC_VIEW_REGISTRY_WORDS = 4 + 4 * SC_VIEW_SLOTS;   // [0] call counter; per slot: forward axis (3 floats), stamp
__host__ __device__ static inline int sc_clamp_view_slot(int v) { return v < 0 ? 0 : (v >= SC_VIEW_SLOTS ? SC_VIEW_SLOTS - 1 : v); }
extern int g_sc_isect_pull;          // sc_set_option "isect_pull" (isect_bin.hip)
extern int g_sc_raster_bwd_split;    // sc_set_option "raster_bwd_split" (raster_bwd.hip)
extern int g_sc_raster_hint_blend;   // sc_set_option "raster_hint_blend"
extern int g_sc_raster_split;   // sc_set_option "raster_split" (raster_fwd.hip; read by the order job of isect_bin.hip)

static inline hipStream_t sc_s(sc_stream_t s) { return (hipStream_t)s; }

static inline size_t sc_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// floor(log2(n)) + 1 for n >= 1  (tile_bits / cam_bits)
static inline int sc_bits_for(int64_t n) {
    int b = 0;
    while (n > 0) { ++b; n >>= 1; }
    return b < 1 ? 1 : b;
}

#ifdef __HIPCC__
__device__ __forceinline__ int sc_lane() { return (int)(threadIdx.x & 63); }

// inclusive wave scan (64 lanes) of an int
__device__ __forceinline__ int sc_wave_incl_scan(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (sc_lane() >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ long long sc_wave_incl_scan64(long long v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        long long t = __shfl_up(v, d, 64);
        if (sc_lane() >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ float sc_wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ unsigned long long sc_lanemask_lt() {
    return (1ull << sc_lane()) - 1ull;
}
#endif
