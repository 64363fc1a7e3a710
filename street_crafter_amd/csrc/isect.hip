// a3/a4: tile intersection (count, scan, emit) and per-tile offset encoding for gfx950.
// Replaces gsplat.rendering.isect_tiles / isect_offset_encode as called at
// street_gaussian/models/street_gaussian_renderer.py:243-253 (semantics: SURVEY.md A.2, A.3).
// Integer/byte work, HBM-bound; all results are bit-exact against oracle/gsplat_oracle.py.
#include "sc_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int BLK = 256;  // gaussians per workgroup in count / emit

struct Rect { int x0, x1, y0, y1; };

// tile rectangle of one projected Gaussian (explicit clamp to [0, tiles]).
__device__ __forceinline__ Rect tile_rect(float mx, float my, int radius, float tile_size,
                                          int tile_width, int tile_height) {
    Rect r;
    if (radius <= 0) { r.x0 = r.x1 = r.y0 = r.y1 = 0; return r; }
    const float tr = (float)radius / tile_size;
    const float tx = mx / tile_size, ty = my / tile_size;
    const float tw = (float)tile_width, th = (float)tile_height;
    r.x0 = (int)fmaxf(fminf(floorf(tx - tr), tw), 0.0f);
    r.x1 = (int)fmaxf(fminf(ceilf(tx + tr), tw), 0.0f);
    r.y0 = (int)fmaxf(fminf(floorf(ty - tr), th), 0.0f);
    r.y1 = (int)fmaxf(fminf(ceilf(ty + tr), th), 0.0f);
    return r;
}

// block-wide exclusive scan of one int per thread (256 threads = 4 waves); returns block total.
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* lds4) {
    const int lane = sc_lane(), wave = threadIdx.x >> 6;
    const int incl = sc_wave_incl_scan(v);
    if (lane == 63) lds4[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < BLK / 64; ++w) {
        const int s = lds4[w];
        if (w < wave) base += s;
        tot += s;
    }
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(BLK) void isect_count_kernel(
    const float* __restrict__ means2d, const int32_t* __restrict__ radii, int64_t CN,
    float tile_size, int tile_width, int tile_height, int32_t* __restrict__ tiles_per_gauss,
    int64_t* __restrict__ block_sums) {
    __shared__ int lds4[BLK / 64];
    const int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
    int cnt = 0;
    if (i < CN) {
        const float2 m = *reinterpret_cast<const float2*>(means2d + i * 2);
        const Rect r = tile_rect(m.x, m.y, radii[i], tile_size, tile_width, tile_height);
        cnt = (r.y1 - r.y0) * (r.x1 - r.x0);
        tiles_per_gauss[i] = cnt;
    }
    int tot;
    (void)block_excl_scan(cnt, &tot, lds4);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = (int64_t)tot;
}

// single workgroup: in-place exclusive scan of nb int64 block sums; writes the grand total.
__global__ __launch_bounds__(1024) void scan_block_sums_kernel(int64_t* __restrict__ sums, int64_t nb,
                                                               int64_t* __restrict__ total) {
    __shared__ long long wave_tot[16];
    __shared__ long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = sc_lane(), wave = threadIdx.x >> 6;
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const long long v = (i < nb) ? sums[i] : 0;
        const long long incl = sc_wave_incl_scan64(v);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        long long pre = carry_s, tot = 0;
        for (int w = 0; w < 16; ++w) {
            const long long s = wave_tot[w];
            if (w < wave) pre += s;
            tot += s;
        }
        if (i < nb) sums[i] = pre + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

// Emit keys/values.  The 256 Gaussians of a workgroup own one contiguous output range; the
// workgroup's threads stride over that range (coalesced 8-B / 4-B stores, perfectly balanced
// whatever the per-Gaussian tile counts are) and find each slot's owner by binary search
// over the in-LDS prefix sums.
__global__ __launch_bounds__(BLK) void isect_emit_kernel(
    const float* __restrict__ means2d, const int32_t* __restrict__ radii,
    const float* __restrict__ depths, int64_t CN, int N, float tile_size, int tile_width,
    int tile_height, int tile_bits, const int64_t* __restrict__ block_offsets,
    int64_t* __restrict__ isect_ids, int32_t* __restrict__ flatten_ids) {
    __shared__ int lds4[BLK / 64];
    __shared__ int pre_s[BLK + 1];
    __shared__ int rx0_s[BLK], rw_s[BLK], ry0_s[BLK];
    __shared__ unsigned int dbits_s[BLK];
    __shared__ int cam_s[BLK];
    const int t = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * BLK + t;
    int cnt = 0, cam = 0;
    Rect r = {0, 0, 0, 0};
    unsigned int db = 0;
    if (i < CN) {
        const float2 m = *reinterpret_cast<const float2*>(means2d + i * 2);
        r = tile_rect(m.x, m.y, radii[i], tile_size, tile_width, tile_height);
        cnt = (r.y1 - r.y0) * (r.x1 - r.x0);
        db = __float_as_uint(depths[i]);
        cam = (int)(i / N);
    }
    int tot;
    const int excl = block_excl_scan(cnt, &tot, lds4);
    pre_s[t] = excl;
    rx0_s[t] = r.x0; rw_s[t] = r.x1 - r.x0; ry0_s[t] = r.y0; dbits_s[t] = db; cam_s[t] = cam;
    if (t == 0) pre_s[BLK] = tot;
    __syncthreads();
    const int64_t out_base = block_offsets[blockIdx.x];
    const int64_t g_base = (int64_t)blockIdx.x * BLK;
    for (int s = t; s < tot; s += BLK) {
        // largest j with pre_s[j] <= s  (owners with zero tiles are skipped automatically)
        int lo = 0, hi = BLK;  // invariant: pre_s[lo] <= s < pre_s[hi]
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int mid = (lo + hi) >> 1;
            if (pre_s[mid] <= s) lo = mid; else hi = mid;
        }
        const int local = s - pre_s[lo];
        const int w = rw_s[lo];
        const int ty = ry0_s[lo] + local / w;
        const int tx = rx0_s[lo] + local % w;
        const int64_t gi = g_base + lo;           // flat (camera, gaussian) index
        const int64_t cid = cam_s[lo];
        const int64_t key = (cid << (32 + tile_bits)) |
                            ((int64_t)(ty * tile_width + tx) << 32) | (int64_t)dbits_s[lo];
        isect_ids[out_base + s] = key;
        flatten_ids[out_base + s] = (int32_t)gi;
    }
}

// a4: offsets[flat_tile] = first index whose (cam, tile) >= flat_tile (lower bound).
__global__ __launch_bounds__(256) void isect_offsets_kernel(
    const int64_t* __restrict__ isect_ids, int64_t I, int n_tiles, int tile_bits, int total_tiles,
    int32_t* __restrict__ offsets) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= I) return;
    const int64_t tile_mask = ((int64_t)1 << tile_bits) - 1;
    const int64_t k = isect_ids[i] >> 32;
    const int64_t cur = (k >> tile_bits) * n_tiles + (k & tile_mask);
    int64_t lo;
    if (i == 0) {
        lo = 0;
    } else {
        const int64_t kp = isect_ids[i - 1] >> 32;
        lo = (kp >> tile_bits) * n_tiles + (kp & tile_mask) + 1;
    }
    for (int64_t tid = lo; tid <= cur && tid < total_tiles; ++tid) offsets[tid] = (int32_t)i;
    if (i == I - 1)
        for (int64_t tid = cur + 1; tid < total_tiles; ++tid) offsets[tid] = (int32_t)I;
}

}  // namespace

static inline int64_t isect_num_blocks(int64_t CN) { return (CN + BLK - 1) / BLK; }

extern "C" size_t sc_isect_workspace_bytes(int64_t CN) {
    return sc_align_up((size_t)(isect_num_blocks(CN) + 1) * sizeof(int64_t), 256);
}

extern "C" int sc_isect_count(const float* means2d, const int32_t* radii, int C, int N, int tile_size,
                              int tile_width, int tile_height, int32_t* tiles_per_gauss,
                              int64_t* total_dev, void* workspace, size_t ws_bytes,
                              sc_stream_t stream) {
    if (C < 0 || N < 0 || tile_size <= 0 || tile_width <= 0 || tile_height <= 0) return SC_EINVAL;
    if (!total_dev) return SC_EINVAL;
    const int64_t CN = (int64_t)C * N;
    if (CN == 0) return (int)hipMemsetAsync(total_dev, 0, sizeof(int64_t), sc_s(stream));
    if (!means2d || !radii || !tiles_per_gauss || !workspace) return SC_EINVAL;
    if (ws_bytes < sc_isect_workspace_bytes(CN)) return SC_EWORKSPACE;
    const int64_t nb = isect_num_blocks(CN);
    if (nb > 0x7fffffff) return SC_EINVAL;
    int64_t* block_sums = (int64_t*)workspace;
    hipLaunchKernelGGL(isect_count_kernel, dim3((unsigned)nb), dim3(BLK), 0, sc_s(stream), means2d,
                       radii, CN, (float)tile_size, tile_width, tile_height, tiles_per_gauss,
                       block_sums);
    SC_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, sc_s(stream), block_sums, nb,
                       total_dev);
    SC_LAUNCH_CHECK();
    return SC_OK;
}

extern "C" int sc_isect_emit(const float* means2d, const int32_t* radii, const float* depths, int C,
                             int N, int tile_size, int tile_width, int tile_height,
                             const int32_t* tiles_per_gauss, int64_t n_isects, int64_t* isect_ids,
                             int32_t* flatten_ids, void* workspace, size_t ws_bytes,
                             sc_stream_t stream) {
    (void)tiles_per_gauss;
    if (C < 0 || N < 0 || tile_size <= 0 || tile_width <= 0 || tile_height <= 0 || n_isects < 0)
        return SC_EINVAL;
    const int64_t CN = (int64_t)C * N;
    if (CN == 0 || n_isects == 0) return SC_OK;
    if (!means2d || !radii || !depths || !isect_ids || !flatten_ids || !workspace) return SC_EINVAL;
    if (ws_bytes < sc_isect_workspace_bytes(CN)) return SC_EWORKSPACE;
    if (CN > 0x7fffffff || n_isects > 0x7fffffffLL) return SC_EINVAL;  // flatten_ids and the offsets into them are int32
    const int64_t nb = isect_num_blocks(CN);
    const int tile_bits = sc_bits_for((int64_t)tile_width * tile_height);
    hipLaunchKernelGGL(isect_emit_kernel, dim3((unsigned)nb), dim3(BLK), 0, sc_s(stream), means2d,
                       radii, depths, CN, N, (float)tile_size, tile_width, tile_height, tile_bits,
                       (const int64_t*)workspace, isect_ids, flatten_ids);
    SC_LAUNCH_CHECK();
    return SC_OK;
}

extern "C" int sc_isect_offsets(const int64_t* isect_ids, int64_t n_isects, int C, int tile_width,
                                int tile_height, int32_t* offsets, sc_stream_t stream) {
    if (C < 0 || tile_width <= 0 || tile_height <= 0 || n_isects < 0) return SC_EINVAL;
    const int64_t total_tiles = (int64_t)C * tile_width * tile_height;
    if (total_tiles == 0) return SC_OK;
    if (!offsets || total_tiles > 0x7fffffff || n_isects > 0x7fffffff) return SC_EINVAL;
    if (n_isects == 0)
        return (int)hipMemsetAsync(offsets, 0, (size_t)total_tiles * sizeof(int32_t), sc_s(stream));
    if (!isect_ids) return SC_EINVAL;
    const int tile_bits = sc_bits_for((int64_t)tile_width * tile_height);
    const int64_t nb = (n_isects + 255) / 256;
    hipLaunchKernelGGL(isect_offsets_kernel, dim3((unsigned)nb), dim3(256), 0, sc_s(stream), isect_ids,
                       n_isects, tile_width * tile_height, tile_bits, (int)total_tiles, offsets);
    SC_LAUNCH_CHECK();
    return SC_OK;
}
