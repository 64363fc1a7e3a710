// a14: simple_knn._C.distCUDA2 for gfx950 -- mean squared distance to the 3 nearest other points.
// Call sites in the reference: street_gaussian/models/gaussian_model.py:65,
// gaussian_model_actor.py:139, data_processor/utils/render_utils.py:125 (semantics: SURVEY.md A.7).
//
// Exact 3-NN.  Points are ordered along a 30-bit Morton curve (radix_sort.hip), cut into boxes of
// 1024 consecutive points with an AABB each; a point first bounds its answer with its +-3
// neighbours on the curve, then visits only boxes whose AABB is nearer than that bound.  Lanes of
// a wave are neighbours on the curve, so they walk almost the same boxes and the box reads are
// broadcast loads.  Distances use the oracle's op order without FMA contraction and the pruning
// test is monotone in fp32, so the output is bit-identical to oracle/knn_oracle.py.
#include "sc_common.h"
#include <float.h>

#pragma clang fp contract(off)

namespace {

constexpr int KNN_BOX = 1024;

struct MinMax { float lo[3], hi[3]; };

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}

// AABB of `count` points starting at pts[start]; one workgroup (256 threads) per box.
// With ids != nullptr the points are gathered through ids (sorted order) and also written,
// gathered, to sorted_pts so that later passes stream them.
__global__ __launch_bounds__(256) void knn_box_minmax_kernel(const float* __restrict__ pts,
                                                             const int* __restrict__ ids, int64_t n,
                                                             int box_size, float* __restrict__ sorted_pts,
                                                             MinMax* __restrict__ boxes) {
    __shared__ float red[6][4];
    const int64_t start = (int64_t)blockIdx.x * box_size;
    const int64_t end = min(start + box_size, n);
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int64_t i = start + threadIdx.x; i < end; i += 256) {
        const int64_t src = ids ? (int64_t)ids[i] : i;
        const float x = pts[src * 3 + 0], y = pts[src * 3 + 1], z = pts[src * 3 + 2];
        if (sorted_pts) { sorted_pts[i * 3 + 0] = x; sorted_pts[i * 3 + 1] = y; sorted_pts[i * 3 + 2] = z; }
        lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
        hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
    }
    const int lane = sc_lane(), wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = wave_min(lo[a]), h = wave_max(hi[a]);
        if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        MinMax m;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            m.lo[a] = fminf(fminf(red[a][0], red[a][1]), fminf(red[a][2], red[a][3]));
            m.hi[a] = fmaxf(fmaxf(red[3 + a][0], red[3 + a][1]), fmaxf(red[3 + a][2], red[3 + a][3]));
        }
        boxes[blockIdx.x] = m;
    }
}

// single workgroup: reduce nb partial AABBs to one
__global__ __launch_bounds__(256) void knn_reduce_boxes_kernel(const MinMax* __restrict__ boxes, int nb,
                                                               MinMax* __restrict__ out) {
    __shared__ float red[6][4];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = threadIdx.x; i < nb; i += 256) {
        const MinMax m = boxes[i];
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], m.lo[a]); hi[a] = fmaxf(hi[a], m.hi[a]); }
    }
    const int lane = sc_lane(), wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = wave_min(lo[a]), h = wave_max(hi[a]);
        if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        MinMax m;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            m.lo[a] = fminf(fminf(red[a][0], red[a][1]), fminf(red[a][2], red[a][3]));
            m.hi[a] = fmaxf(fmaxf(red[3 + a][0], red[3 + a][1]), fmaxf(red[3 + a][2], red[3 + a][3]));
        }
        *out = m;
    }
}

__device__ __forceinline__ unsigned spread10(unsigned x) {
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

__global__ __launch_bounds__(256) void knn_morton_kernel(const float* __restrict__ pts, int64_t n,
                                                         const MinMax* __restrict__ bbox,
                                                         unsigned long long* __restrict__ keys,
                                                         int* __restrict__ ids) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const MinMax m = *bbox;
    unsigned q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float ext = m.hi[a] - m.lo[a];
        float r = ext > 0.f ? (pts[i * 3 + a] - m.lo[a]) / ext : 0.f;
        r = fminf(fmaxf(r, 0.f), 1.f);      // NaN -> 0
        q[a] = (unsigned)(r * 1023.0f);
    }
    keys[i] = (unsigned long long)(spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2));
    ids[i] = (int)i;
}

__device__ __forceinline__ void knn_insert(float d, float& b0, float& b1, float& b2) {
    if (d < b2) {
        if (d < b1) {
            b2 = b1;
            if (d < b0) { b1 = b0; b0 = d; } else { b1 = d; }
        } else {
            b2 = d;
        }
    }
}

__device__ __forceinline__ float dist2(float px, float py, float pz, const float* __restrict__ q) {
    const float dx = px - q[0], dy = py - q[1], dz = pz - q[2];
    return (dx * dx + dy * dy) + dz * dz;
}

__device__ __forceinline__ float box_dist2(float px, float py, float pz, const MinMax& m) {
    // per-axis gap; 0 inside.  (p - lo) / (p - hi) use the same subtraction order as dist2 so the
    // bound is monotone in fp32: for any q in the box, dist2(p,q) >= box_dist2(p,box) exactly.
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (px < m.lo[0]) dx = px - m.lo[0]; else if (px > m.hi[0]) dx = px - m.hi[0];
    if (py < m.lo[1]) dy = py - m.lo[1]; else if (py > m.hi[1]) dy = py - m.hi[1];
    if (pz < m.lo[2]) dz = pz - m.lo[2]; else if (pz > m.hi[2]) dz = pz - m.hi[2];
    return (dx * dx + dy * dy) + dz * dz;
}

__global__ __launch_bounds__(256) void knn_mean_dist_kernel(const float* __restrict__ sorted_pts,
                                                            const int* __restrict__ ids, int64_t n,
                                                            const MinMax* __restrict__ boxes, int nb,
                                                            float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float px = sorted_pts[i * 3 + 0], py = sorted_pts[i * 3 + 1], pz = sorted_pts[i * 3 + 2];
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    for (int64_t j = max((int64_t)0, i - 3); j <= min(n - 1, i + 3); ++j) {
        if (j == i) continue;
        knn_insert(dist2(px, py, pz, sorted_pts + j * 3), b0, b1, b2);
    }
    const float reject = b2;
    b0 = b1 = b2 = FLT_MAX;
    for (int b = 0; b < nb; ++b) {
        const MinMax m = boxes[b];
        const float bd = box_dist2(px, py, pz, m);
        if (bd > reject || bd > b2) continue;
        const int64_t s = (int64_t)b * KNN_BOX, e = min(s + KNN_BOX, n);
        for (int64_t j = s; j < e; ++j) {
            if (j == i) continue;
            knn_insert(dist2(px, py, pz, sorted_pts + j * 3), b0, b1, b2);
        }
    }
    out[ids[i]] = ((b0 + b1) + b2) / 3.0f;
}

}  // namespace

extern "C" size_t sc_radix_sort_workspace_bytes(int64_t n);
extern "C" int sc_radix_sort_pairs_u64_i32(uint64_t*, int32_t*, uint64_t*, int32_t*, int64_t, int, void*,
                                           size_t, sc_stream_t);

namespace {
struct KnnLayout {
    size_t keys, tmp_keys, ids, tmp_ids, sorted, boxes, bbox, sort_ws, total;
};
KnnLayout knn_layout(int64_t n) {
    KnnLayout L;
    const size_t nb = (size_t)((n + KNN_BOX - 1) / KNN_BOX) + 1;
    size_t o = 0;
    L.keys = o; o += sc_align_up((size_t)n * 8, 256);
    L.tmp_keys = o; o += sc_align_up((size_t)n * 8, 256);
    L.ids = o; o += sc_align_up((size_t)n * 4, 256);
    L.tmp_ids = o; o += sc_align_up((size_t)n * 4, 256);
    L.sorted = o; o += sc_align_up((size_t)n * 12, 256);
    L.boxes = o; o += sc_align_up(nb * sizeof(MinMax), 256);
    L.bbox = o; o += 256;
    L.sort_ws = o; o += sc_radix_sort_workspace_bytes(n);
    L.total = o;
    return L;
}
}  // namespace

extern "C" size_t sc_knn_workspace_bytes(int64_t n) {
    if (n <= 0) return 256;
    return knn_layout(n).total;
}

extern "C" int sc_knn3_mean_dist2(const float* points, int64_t n, float* out, void* workspace,
                                  size_t ws_bytes, sc_stream_t stream) {
    if (n < 0) return SC_EINVAL;
    if (n == 0) return SC_OK;
    if (n > 0x7fffffffLL) return SC_EINVAL;
    if (!points || !out || !workspace) return SC_EINVAL;
    const KnnLayout L = knn_layout(n);
    if (ws_bytes < L.total) return SC_EWORKSPACE;
    unsigned char* ws = (unsigned char*)workspace;
    unsigned long long* keys = (unsigned long long*)(ws + L.keys);
    int* ids = (int*)(ws + L.ids);
    float* sorted = (float*)(ws + L.sorted);
    MinMax* boxes = (MinMax*)(ws + L.boxes);
    MinMax* bbox = (MinMax*)(ws + L.bbox);
    const int nb = (int)((n + KNN_BOX - 1) / KNN_BOX);
    hipStream_t s = sc_s(stream);
    // 1. global bounding box (unsorted boxes -> one)
    hipLaunchKernelGGL(knn_box_minmax_kernel, dim3(nb), dim3(256), 0, s, points, (const int*)nullptr, n,
                       KNN_BOX, (float*)nullptr, boxes);
    SC_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_reduce_boxes_kernel, dim3(1), dim3(256), 0, s, boxes, nb, bbox);
    SC_LAUNCH_CHECK();
    // 2. Morton codes, 3. sort
    hipLaunchKernelGGL(knn_morton_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, points, n,
                       bbox, keys, ids);
    SC_LAUNCH_CHECK();
    int rc = sc_radix_sort_pairs_u64_i32((uint64_t*)keys, ids, (uint64_t*)(ws + L.tmp_keys),
                                         (int*)(ws + L.tmp_ids), n, 30, ws + L.sort_ws,
                                         sc_radix_sort_workspace_bytes(n), stream);
    if (rc != SC_OK) return rc;
    // 4. gather into curve order + per-box AABBs
    hipLaunchKernelGGL(knn_box_minmax_kernel, dim3(nb), dim3(256), 0, s, points, (const int*)ids, n, KNN_BOX,
                       sorted, boxes);
    SC_LAUNCH_CHECK();
    // 5. pruned exact scan
    hipLaunchKernelGGL(knn_mean_dist_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sorted,
                       (const int*)ids, n, boxes, nb, out);
    SC_LAUNCH_CHECK();
    return SC_OK;
}
