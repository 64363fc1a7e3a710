// Tile-bucketed intersection path for gfx950: the fast route behind
// gsplat.rendering.isect_tiles(sort=True) + isect_offset_encode
// (street_gaussian/models/street_gaussian_renderer.py:243-253).
//
// The reference-shaped route (isect.hip + radix_sort.hip) moves every (key, value) pair through
// HBM once per radix pass: 6 x 24 B x I.  The sort key is (camera, tile, depth bits), and the
// number of (camera, tile) buckets is small (9600 at 1920x1280), so instead:
//   1. bin_count      : per-workgroup LDS histograms -> global per-bucket counts -> exclusive scan
//                       = isect_offsets (exactly isect_offset_encode's lower bounds), total, max.
//                       The same pass histograms the visible Gaussians by the tile of their
//                       CENTRE and a small counting sort (center_scatter) orders them spatially.
//   2. bin_scatter    : walks the Gaussians in that spatial order, so the 4096 Gaussians of a
//                       workgroup touch a compact strip of buckets: it reserves a slice of each
//                       bucket it touches with ONE global atomic per (workgroup, bucket) and drops
//                       8-byte (depth bits, flat id) records into its slices; records of one bucket
//                       arrive in long runs, which is what makes the stores coalesce (in arrival
//                       order of the Gaussians the same stores ran ~3x slower).
//   3. tile_bucket_sort: one workgroup per bucket: interpolation sort in LDS on (depth bits, id);
//                       degenerate tiles (many equal depths) fall back to tile_sort (ballot-ranked
//                       stable LSD radix passes).
// The required order is (depth bits, flat id): a stable sort of gaussian-major emission order
// breaks depth ties by ascending flat id.  Integer work only: results are bit-identical to the
// reference-shaped route (tests compare both against the oracle).
// HBM traffic: ~(8 + 8 + 12) B x I instead of ~144 B x I.
#include "sc_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int BIN_THREADS = 1024;
constexpr int BIN_GPT = 4;                       // gaussians per thread
constexpr int BIN_GPB = BIN_THREADS * BIN_GPT;   // gaussians per workgroup
constexpr int BIN_MAX_BUCKETS = 16384;           // (camera, tile) buckets that fit the LDS histograms
constexpr int BIN_BIG = 64;                      // rectangles larger than this are walked by a whole wave

struct Rect { int x0, x1, y0, y1; };

// identical to isect.hip's tile_rect
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

// Calls f(bucket, pa, pb) once for every tile of this lane's rectangle, where (pa, pb) is the
// owning lane's payload.  Rectangles with more than BIN_BIG tiles are spread over the 64 lanes of
// the wave (the payload is broadcast while the wave is still converged).  All lanes of a wave
// must call this together.
template <typename F>
__device__ __forceinline__ void walk_rect(const Rect& r, int cnt, int bucket_base, int tile_width,
                                          unsigned pa, unsigned pb, F&& f) {
    const int w = r.x1 - r.x0;
    if (cnt > 0 && cnt <= BIN_BIG) {
        for (int ty = r.y0; ty < r.y1; ++ty) {
            const int row = bucket_base + ty * tile_width;
            for (int tx = r.x0; tx < r.x1; ++tx) f(row + tx, pa, pb);
        }
    }
    unsigned long long big = __ballot(cnt > BIN_BIG);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const int bx0 = __shfl(r.x0, src, 64), by0 = __shfl(r.y0, src, 64);
        const int bw = __shfl(w, src, 64), bcnt = __shfl(cnt, src, 64);
        const int bbase = __shfl(bucket_base, src, 64);
        const unsigned ba = (unsigned)__shfl((int)pa, src, 64), bb = (unsigned)__shfl((int)pb, src, 64);
        for (int s = sc_lane(); s < bcnt; s += 64) {
            const int ty = by0 + s / bw, tx = bx0 + s % bw;
            f(bbase + ty * tile_width + tx, ba, bb);
        }
    }
}

// ---- pass 1: counts -----------------------------------------------------------------------------
// counts[b]  += number of rectangles covering bucket b          (-> isect_offsets)
// ccounts[b] += number of visible Gaussians whose CENTRE tile is b (-> spatial order)
__global__ __launch_bounds__(BIN_THREADS) void bin_count_kernel(
    const float* __restrict__ means2d, const int32_t* __restrict__ radii, int64_t CN, int N,
    float tile_size, int tile_width, int tile_height, int n_buckets,
    int32_t* __restrict__ tiles_per_gauss, unsigned* __restrict__ counts,
    unsigned* __restrict__ ccounts) {
    extern __shared__ unsigned lds[];
    unsigned* hist = lds;                 // [n_buckets]
    unsigned* chist = lds + n_buckets;    // [n_buckets]
    for (int b = threadIdx.x; b < 2 * n_buckets; b += BIN_THREADS) lds[b] = 0;
    __syncthreads();
    const int T = tile_width * tile_height;
    const int64_t base = (int64_t)blockIdx.x * BIN_GPB;
#pragma unroll
    for (int k = 0; k < BIN_GPT; ++k) {
        const int64_t i = base + (int64_t)k * BIN_THREADS + threadIdx.x;
        Rect r = {0, 0, 0, 0};
        int cnt = 0, bbase = 0;
        if (i < CN) {
            const float2 m = *reinterpret_cast<const float2*>(means2d + i * 2);
            r = tile_rect(m.x, m.y, radii[i], tile_size, tile_width, tile_height);
            cnt = (r.y1 - r.y0) * (r.x1 - r.x0);
            tiles_per_gauss[i] = cnt;
            bbase = (int)(i / N) * T;
            if (cnt > 0) {
                // centre of the (non-empty) rectangle: a tile the Gaussian is filed under
                const int cx = (r.x0 + r.x1 - 1) >> 1, cy = (r.y0 + r.y1 - 1) >> 1;
                atomicAdd(&chist[bbase + cy * tile_width + cx], 1u);
            }
        }
        walk_rect(r, cnt, bbase, tile_width, 0u, 0u,
                  [&](int bucket, unsigned, unsigned) { atomicAdd(&hist[bucket], 1u); });
    }
    __syncthreads();
    for (int b = threadIdx.x; b < n_buckets; b += BIN_THREADS) {
        const unsigned c = hist[b], cc = chist[b];
        if (c) atomicAdd(&counts[b], c);
        if (cc) atomicAdd(&ccounts[b], cc);
    }
}

// single workgroup: out = exclusive scan of counts; meta[0] = total, meta[1] = max count.
// Thread t owns `per` consecutive counters (one pass, one block scan).
__global__ __launch_bounds__(1024) void bin_scan_kernel(const unsigned* __restrict__ counts, int n_buckets,
                                                        int32_t* __restrict__ out,
                                                        int64_t* __restrict__ meta) {
    __shared__ long long wave_tot[16];
    __shared__ unsigned wave_max[16];
    const int t = threadIdx.x, lane = sc_lane(), wave = t >> 6;
    const int per = (n_buckets + 1023) / 1024;
    const int beg = t * per, end = min(beg + per, n_buckets);
    long long sum = 0;
    unsigned mx = 0;
    for (int i = beg; i < end; ++i) {
        const unsigned c = counts[i];
        sum += c;
        mx = max(mx, c);
    }
    const long long incl = sc_wave_incl_scan64(sum);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o, 64));
    if (lane == 63) wave_tot[wave] = incl;
    if (lane == 0) wave_max[wave] = mx;
    __syncthreads();
    long long run = incl - sum, tot = 0;
    unsigned m = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        if (w < wave) run += wave_tot[w];
        tot += wave_tot[w];
        m = max(m, wave_max[w]);
    }
    for (int i = beg; i < end; ++i) {
        const unsigned c = counts[i];
        out[i] = (int32_t)run;
        run += c;
    }
    if (t == 0) { meta[0] = tot; meta[1] = (long long)m; }
}

// ---- spatial order: counting sort of the visible Gaussians by centre tile -------------------------
__global__ __launch_bounds__(BIN_THREADS) void center_scatter_kernel(
    const int32_t* __restrict__ tiles_per_gauss, const float* __restrict__ means2d,
    const int32_t* __restrict__ radii, int64_t CN, int N, float tile_size, int tile_width,
    int tile_height, int n_buckets, const int32_t* __restrict__ cstart, unsigned* __restrict__ ccursor,
    int32_t* __restrict__ perm) {
    extern __shared__ unsigned lds[];
    unsigned* hist = lds;               // [n_buckets]
    unsigned* gbase = lds + n_buckets;  // [n_buckets]
    for (int b = threadIdx.x; b < n_buckets; b += BIN_THREADS) hist[b] = 0;
    __syncthreads();
    const int T = tile_width * tile_height;
    const int64_t base = (int64_t)blockIdx.x * BIN_GPB;
    int cb[BIN_GPT];
#pragma unroll
    for (int k = 0; k < BIN_GPT; ++k) {
        const int64_t i = base + (int64_t)k * BIN_THREADS + threadIdx.x;
        cb[k] = -1;
        if (i < CN && tiles_per_gauss[i] > 0) {
            const float2 m = *reinterpret_cast<const float2*>(means2d + i * 2);
            const Rect r = tile_rect(m.x, m.y, radii[i], tile_size, tile_width, tile_height);
            const int cx = (r.x0 + r.x1 - 1) >> 1, cy = (r.y0 + r.y1 - 1) >> 1;
            cb[k] = (int)(i / N) * T + cy * tile_width + cx;
            atomicAdd(&hist[cb[k]], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < n_buckets; b += BIN_THREADS) {
        const unsigned c = hist[b];
        if (c) gbase[b] = (unsigned)cstart[b] + atomicAdd(&ccursor[b], c);
        hist[b] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BIN_GPT; ++k) {
        const int64_t i = base + (int64_t)k * BIN_THREADS + threadIdx.x;
        if (cb[k] >= 0) perm[gbase[cb[k]] + atomicAdd(&hist[cb[k]], 1u)] = (int32_t)i;
    }
}

// ---- pass 2: records ------------------------------------------------------------------------------
__global__ __launch_bounds__(BIN_THREADS) void bin_scatter_kernel(
    const float* __restrict__ means2d, const int32_t* __restrict__ radii,
    const float* __restrict__ depths, const int32_t* __restrict__ perm,
    const int64_t* __restrict__ n_visible, int N, float tile_size, int tile_width,
    int tile_height, int n_buckets, const int32_t* __restrict__ offsets,
    const int64_t* __restrict__ meta, int64_t capacity, int64_t tile_capacity,
    unsigned* __restrict__ cursor, uint2* __restrict__ bucket, int dbg) {
    extern __shared__ unsigned lds[];
    // the caller may have sized the buffers from a prediction: do nothing if they are too small
    if (meta[0] > capacity || meta[1] > tile_capacity) return;
    const int64_t M = n_visible[0];
    const int64_t base = (int64_t)blockIdx.x * BIN_GPB;
    if (base >= M) return;
    unsigned* hist = lds;               // [n_buckets] counts, then running local cursors
    unsigned* gbase = lds + n_buckets;  // [n_buckets] global start of this workgroup's slice
    for (int b = threadIdx.x; b < n_buckets; b += BIN_THREADS) hist[b] = 0;
    __syncthreads();
    const int T = tile_width * tile_height;
    Rect rr[BIN_GPT];
    int cc[BIN_GPT], bb[BIN_GPT];
    unsigned dd[BIN_GPT], ii[BIN_GPT];
#pragma unroll
    for (int k = 0; k < BIN_GPT; ++k) {
        // consecutive lanes take consecutive Gaussians of the spatial order
        const int64_t j = base + (int64_t)k * BIN_THREADS + threadIdx.x;
        rr[k] = {0, 0, 0, 0};
        cc[k] = 0; bb[k] = 0; dd[k] = 0; ii[k] = 0;
        if (j < M) {
            const int64_t i = perm[j];
            const float2 m = *reinterpret_cast<const float2*>(means2d + i * 2);
            rr[k] = tile_rect(m.x, m.y, radii[i], tile_size, tile_width, tile_height);
            cc[k] = (rr[k].y1 - rr[k].y0) * (rr[k].x1 - rr[k].x0);
            bb[k] = (int)(i / N) * T;
            dd[k] = __float_as_uint(depths[i]);
            ii[k] = (unsigned)i;
        }
        walk_rect(rr[k], cc[k], bb[k], tile_width, 0u, 0u,
                  [&](int b, unsigned, unsigned) { atomicAdd(&hist[b], 1u); });
    }
    __syncthreads();
    for (int b = threadIdx.x; b < n_buckets; b += BIN_THREADS) {
        const unsigned c = hist[b];
        if (c) gbase[b] = (unsigned)offsets[b] + ((dbg & 4) ? 0u : atomicAdd(&cursor[b], c));
        hist[b] = 0;
    }
    __syncthreads();
    if (dbg & 2) return;
#pragma unroll
    for (int k = 0; k < BIN_GPT; ++k) {
        walk_rect(rr[k], cc[k], bb[k], tile_width, dd[k], ii[k], [&](int b, unsigned d, unsigned id) {
            const unsigned slot = gbase[b] + atomicAdd(&hist[b], 1u);
            if (!(dbg & 1)) bucket[slot] = make_uint2(d, id);
        });
    }
}

// ---- per-tile LDS sort: radix fallback -------------------------------------------------------------
constexpr int TS_THREADS = 256;
constexpr int TS_WAVES = 4;

__device__ __forceinline__ unsigned long long ts_match(unsigned d, int bits, bool valid) {
    unsigned long long peers = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

// one stable LSD pass over `bits` (<= 8) key bits starting at `shift`: src -> dst, both in LDS
__device__ __forceinline__ void ts_pass(const unsigned long long* __restrict__ src,
                                        unsigned long long* __restrict__ dst, int n, int chunk,
                                        int shift, int bits, unsigned (*h)[256], unsigned* wtot) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned mask = (1u << bits) - 1u;
#pragma unroll
    for (int w = 0; w < TS_WAVES; ++w) h[w][t] = 0;
    __syncthreads();
    const int wbeg = wave * chunk, wend = min(wbeg + chunk, n);
    for (int i = wbeg + lane; i < wend; i += 64)
        atomicAdd(&h[wave][(unsigned)(src[i] >> shift) & mask], 1u);
    __syncthreads();
    {
        unsigned c[TS_WAVES], tot = 0;
#pragma unroll
        for (int w = 0; w < TS_WAVES; ++w) { c[w] = h[w][t]; tot += c[w]; }
        const unsigned incl = (unsigned)sc_wave_incl_scan((int)tot);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        unsigned run = incl - tot;
#pragma unroll
        for (int w = 0; w < TS_WAVES; ++w) if (w < wave) run += wtot[w];
#pragma unroll
        for (int w = 0; w < TS_WAVES; ++w) { h[w][t] = run; run += c[w]; }
    }
    __syncthreads();
    for (int i0 = wbeg; i0 < wend; i0 += 64) {
        const int i = i0 + lane;
        const bool valid = i < wend;
        const unsigned long long k = valid ? src[i] : 0ull;
        const unsigned d = (unsigned)(k >> shift) & mask;
        const unsigned long long peers = ts_match(d, bits, valid);
        const unsigned rank = (unsigned)__popcll(peers & sc_lanemask_lt());
        unsigned pos = 0;
        if (valid) pos = h[wave][d] + rank;
        if (valid && rank == 0) h[wave][d] += (unsigned)__popcll(peers);
        if (valid) dst[pos] = k;
    }
    __syncthreads();
}

__global__ __launch_bounds__(TS_THREADS) void tile_sort_kernel(
    const uint2* __restrict__ bucket, const int32_t* __restrict__ offsets, int n_buckets,
    const int64_t* __restrict__ meta, int64_t capacity, int tiles_per_cam, int tile_bits, int id_bits, int cap,
    const unsigned char* __restrict__ needs_radix, int64_t* __restrict__ isect_ids,
    int32_t* __restrict__ flatten_ids) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long* A = reinterpret_cast<unsigned long long*>(smem);
    unsigned long long* B = A + cap;
    __shared__ unsigned h[TS_WAVES][256];
    __shared__ unsigned wtot[TS_WAVES];
    __shared__ unsigned diff_s;
    __shared__ int tie_s;

    if (meta[0] > capacity || meta[1] > (int64_t)cap) return;    // undersized prediction: caller retries
    const int b = blockIdx.x;
    if (needs_radix && !needs_radix[b]) return;                  // already sorted by tile_bucket_sort_kernel
    const int s = offsets[b];
    const int e = (b + 1 < n_buckets) ? offsets[b + 1] : (int)meta[0];
    const int n = e - s;
    if (n <= 0) return;
    const int t = threadIdx.x;
    if (t == 0) { diff_s = 0; tie_s = 0; }
    __syncthreads();
    const unsigned first_depth = bucket[s].x;
    unsigned diff = 0;
    for (int i = t; i < n; i += TS_THREADS) {
        const uint2 r = bucket[s + i];
        A[i] = ((unsigned long long)r.x << 32) | r.y;
        diff |= r.x ^ first_depth;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) diff |= (unsigned)__shfl_xor((int)diff, o, 64);
    if ((t & 63) == 0 && diff) atomicOr(&diff_s, diff);
    __syncthreads();
    const int sig = 32 - __clz((int)diff_s);          // significant depth bits (0 when all equal)
    int chunk = (n + TS_WAVES - 1) / TS_WAVES;
    chunk = (chunk + 63) & ~63;

    unsigned long long* src = A;
    unsigned long long* dst = B;
    // stable LSD passes over the significant depth bits only ...
    for (int done = 0; done < sig;) {
        const int bits = min(8, sig - done);
        ts_pass(src, dst, n, chunk, 32 + done, bits, h, wtot);
        unsigned long long* tmp = src; src = dst; dst = tmp;
        done += bits;
    }
    // ... any two neighbours with the same depth bits? (ties must be ordered by flat id)
    int tie = 0;
    for (int i = t; i + 1 < n; i += TS_THREADS)
        tie |= ((unsigned)(src[i] >> 32) == (unsigned)(src[i + 1] >> 32));
    if (tie) tie_s = 1;
    __syncthreads();
    if (tie_s) {
        // full (depth, id) key.  LSD: id bits first, then the depth bits again.
        for (int done = 0; done < id_bits;) {
            const int bits = min(8, id_bits - done);
            ts_pass(src, dst, n, chunk, done, bits, h, wtot);
            unsigned long long* tmp = src; src = dst; dst = tmp;
            done += bits;
        }
        for (int done = 0; done < sig;) {
            const int bits = min(8, sig - done);
            ts_pass(src, dst, n, chunk, 32 + done, bits, h, wtot);
            unsigned long long* tmp = src; src = dst; dst = tmp;
            done += bits;
        }
    }
    const long long cam = b / tiles_per_cam, tile = b % tiles_per_cam;
    const long long hi = (cam << (32 + tile_bits)) | (tile << 32);
    for (int i = t; i < n; i += TS_THREADS) {
        const unsigned long long k = src[i];
        if (isect_ids) isect_ids[s + i] = hi | (long long)(k >> 32);
        flatten_ids[s + i] = (int32_t)(unsigned)k;
    }
}

// ---- per-tile interpolation (bucket) sort: the common path ---------------------------------------
// The depth keys of one tile are spread over 2n sub-buckets by a MONOTONE map of their bit
// pattern (so sub-bucket order == key order for any input, NaN and negative patterns included);
// one counting pass groups the records by sub-bucket, then every record finds its final rank by
// counting the smaller (depth, id) keys inside its own sub-bucket (expected occupancy < 1).  One
// histogram + one scan + one scatter + one short rank loop instead of 4-7 ballot-ranked radix
// passes.  A tile whose keys pile up in one sub-bucket (occupancy > BS_MAX_OCC, e.g. hundreds of
// equal depths) is flagged and left to tile_sort_kernel.
constexpr int BS_MAX_OCC = 48;

// E > 0: every thread keeps its <= E records in registers (one global read per record, tiles up to
// 256*E records); E == 0: records are re-read from global/L2 in each of the three passes.
template <int E>
__global__ __launch_bounds__(TS_THREADS) void tile_bucket_sort_kernel(
    const uint2* __restrict__ bucket, const int32_t* __restrict__ offsets, int n_buckets,
    const int64_t* __restrict__ meta, int64_t capacity, int tiles_per_cam, int tile_bits, int cap,
    unsigned char* __restrict__ needs_radix, int64_t* __restrict__ isect_ids,
    int32_t* __restrict__ flatten_ids, int dbg) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long* B = reinterpret_cast<unsigned long long*>(smem);      // [cap]
    unsigned* boff = reinterpret_cast<unsigned*>(smem + (size_t)cap * 8);     // [2*cap + 1]
    __shared__ unsigned red_lo[TS_WAVES], red_hi[TS_WAVES], red_sum[TS_WAVES], red_occ[TS_WAVES];

    if (meta[0] > capacity || meta[1] > (int64_t)cap) return;    // undersized prediction: caller retries
    const int b = blockIdx.x;
    const int s = offsets[b];
    const int e = (b + 1 < n_buckets) ? offsets[b + 1] : (int)meta[0];
    const int n = e - s;
    if (n <= 0) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long long cam = b / tiles_per_cam, tile = b % tiles_per_cam;
    const long long hi_key = (cam << (32 + tile_bits)) | (tile << 32);

    // number of sub-buckets: 2n rounded up to a multiple of 256 (fits: n <= cap, boff holds 2*cap+1)
    const int nbk = ((2 * n + TS_THREADS - 1) / TS_THREADS) * TS_THREADS;
    const int per_thread = nbk / TS_THREADS;
    for (int i = t; i <= nbk; i += TS_THREADS) boff[i] = 0;

    constexpr int EE = E > 0 ? E : 1;
    uint2 rec[EE];
    unsigned lo = 0xffffffffu, hi = 0u;
    if constexpr (E > 0) {
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const int i = t + k * TS_THREADS;
            rec[k] = make_uint2(0u, 0u);
            if (i < n) {
                rec[k] = bucket[s + i];
                lo = min(lo, rec[k].x);
                hi = max(hi, rec[k].x);
            }
        }
    } else {
        for (int i = t; i < n; i += TS_THREADS) {
            const unsigned d = bucket[s + i].x;
            lo = min(lo, d);
            hi = max(hi, d);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, o, 64));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, o, 64));
    }
    if (lane == 0) { red_lo[wave] = lo; red_hi[wave] = hi; }
    __syncthreads();
    lo = min(min(red_lo[0], red_lo[1]), min(red_lo[2], red_lo[3]));
    hi = max(max(red_hi[0], red_hi[1]), max(red_hi[2], red_hi[3]));
    const float scale = (float)nbk / ((float)(hi - lo) + 1.0f);
    auto sub_bucket = [&](unsigned d) -> int {
        const int v = (int)((float)(d - lo) * scale);     // monotone in d
        return min(v, nbk - 1);
    };

    // histogram
    if constexpr (E > 0) {
#pragma unroll
        for (int k = 0; k < E; ++k)
            if (t + k * TS_THREADS < n) atomicAdd(&boff[sub_bucket(rec[k].x)], 1u);
    } else {
        for (int i = t; i < n; i += TS_THREADS) atomicAdd(&boff[sub_bucket(bucket[s + i].x)], 1u);
    }
    __syncthreads();
    // exclusive scan over nbk counters: thread t owns per_thread consecutive counters
    unsigned sum = 0, occ = 0;
    for (int k = 0; k < per_thread; ++k) {
        const unsigned c = boff[t * per_thread + k];
        sum += c;
        occ = max(occ, c);
    }
    const unsigned incl = (unsigned)sc_wave_incl_scan((int)sum);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) occ = max(occ, (unsigned)__shfl_xor((int)occ, o, 64));
    if (lane == 63) red_sum[wave] = incl;
    if (lane == 0) red_occ[wave] = occ;
    __syncthreads();
    occ = max(max(red_occ[0], red_occ[1]), max(red_occ[2], red_occ[3]));
    if (occ > BS_MAX_OCC) {                     // workgroup-uniform
        if (t == 0) needs_radix[b] = 1;
        return;
    }
    unsigned run = incl - sum;
#pragma unroll
    for (int w = 0; w < TS_WAVES; ++w) if (w < wave) run += red_sum[w];
    for (int k = 0; k < per_thread; ++k) {      // counts -> sub-bucket starts (used as running cursors)
        const unsigned c = boff[t * per_thread + k];
        boff[t * per_thread + k] = run;
        run += c;
    }
    __syncthreads();
    // scatter: after this pass boff[j] is the END of sub-bucket j (== start of j+1)
    if constexpr (E > 0) {
#pragma unroll
        for (int k = 0; k < E; ++k) {
            if (t + k * TS_THREADS < n) {
                const unsigned slot = atomicAdd(&boff[sub_bucket(rec[k].x)], 1u);
                B[slot] = ((unsigned long long)rec[k].x << 32) | rec[k].y;
            }
        }
    } else {
        for (int i = t; i < n; i += TS_THREADS) {
            const uint2 r = bucket[s + i];
            const unsigned slot = atomicAdd(&boff[sub_bucket(r.x)], 1u);
            B[slot] = ((unsigned long long)r.x << 32) | r.y;
        }
    }
    __syncthreads();
    // rank inside the sub-bucket by (depth bits, flat id) and write the final records
    if (dbg & 2) return;
    for (int p = t; p < n; p += TS_THREADS) {
        const unsigned long long key = B[p];
        const int j = sub_bucket((unsigned)(key >> 32));
        const unsigned beg = j > 0 ? boff[j - 1] : 0u, end = boff[j];
        unsigned r = beg;
        for (unsigned q = beg; q < end; ++q) r += (B[q] < key) ? 1u : 0u;
        if (dbg & 1) { if (r == 0xffffffffu) flatten_ids[s] = 0; continue; }
        if (isect_ids) isect_ids[s + r] = hi_key | (long long)(key >> 32);
        flatten_ids[s + r] = (int32_t)(unsigned)key;
    }
}

}  // namespace

// ---- host side --------------------------------------------------------------------------------
// count-phase workspace (must be handed to BOTH calls):
//   counts[nb] | ccounts[nb] | ccursor[nb] | cstart[nb] (i32) | cmeta[2] (i64) | perm[CN] (i32)
// sort-phase workspace:
//   cursor[nb] | needs_radix[nb bytes, padded] | records uint2[capacity]
static inline size_t bin_counts_bytes(int n_buckets) { return sc_align_up((size_t)n_buckets * 4, 256); }

struct BinCountLayout { size_t counts, ccounts, ccursor, cstart, cmeta, perm, total; };
static BinCountLayout bin_count_layout(int64_t CN, int nb) {
    BinCountLayout L;
    const size_t cb = bin_counts_bytes(nb);
    L.counts = 0; L.ccounts = cb; L.ccursor = 2 * cb; L.cstart = 3 * cb; L.cmeta = 4 * cb;
    L.perm = 4 * cb + 256;
    L.total = L.perm + sc_align_up((size_t)(CN > 0 ? CN : 0) * 4, 256);
    return L;
}

extern "C" size_t sc_isect_bin_workspace_bytes(int64_t CN, int C, int tile_width, int tile_height,
                                               int64_t n_isects) {
    const int64_t nb = (int64_t)C * tile_width * tile_height;
    if (nb <= 0 || nb > BIN_MAX_BUCKETS) return 256;
    if (n_isects < 0) return bin_count_layout(CN, (int)nb).total;    // count-phase workspace
    return 2 * bin_counts_bytes((int)nb) + sc_align_up((size_t)n_isects * 8, 256) + 256;
}

extern "C" int sc_isect_bin_count(const float* means2d, const int32_t* radii, int C, int N, int tile_size,
                                  int tile_width, int tile_height, int32_t* tiles_per_gauss,
                                  int32_t* isect_offsets, int64_t* meta_dev, void* count_workspace,
                                  size_t ws_bytes, sc_stream_t stream) {
    if (C < 0 || N < 0 || tile_size <= 0 || tile_width <= 0 || tile_height <= 0) return SC_EINVAL;
    if (!meta_dev) return SC_EINVAL;
    const int64_t CN = (int64_t)C * N;
    const int64_t nb64 = (int64_t)C * tile_width * tile_height;
    if (nb64 > BIN_MAX_BUCKETS || CN > 0x7fffffffLL) return SC_EUNSUPPORTED;
    const int nb = (int)nb64;
    hipStream_t s = sc_s(stream);
    if (CN == 0 || nb == 0) {
        if (nb > 0 && isect_offsets) SC_HIP(hipMemsetAsync(isect_offsets, 0, (size_t)nb * 4, s));
        return (int)hipMemsetAsync(meta_dev, 0, 2 * sizeof(int64_t), s);
    }
    if (!means2d || !radii || !tiles_per_gauss || !isect_offsets || !count_workspace) return SC_EINVAL;
    const BinCountLayout L = bin_count_layout(CN, nb);
    if (ws_bytes < L.total) return SC_EWORKSPACE;
    unsigned char* ws = (unsigned char*)count_workspace;
    unsigned* counts = (unsigned*)(ws + L.counts);
    unsigned* ccounts = (unsigned*)(ws + L.ccounts);
    unsigned* ccursor = (unsigned*)(ws + L.ccursor);
    int32_t* cstart = (int32_t*)(ws + L.cstart);
    int64_t* cmeta = (int64_t*)(ws + L.cmeta);
    int32_t* perm = (int32_t*)(ws + L.perm);
    SC_HIP(hipMemsetAsync(ws, 0, 3 * bin_counts_bytes(nb), s));        // counts, ccounts, ccursor
    const unsigned grid = (unsigned)((CN + BIN_GPB - 1) / BIN_GPB);
    hipLaunchKernelGGL(bin_count_kernel, dim3(grid), dim3(BIN_THREADS), (size_t)nb * 8, s, means2d, radii, CN, N,
                       (float)tile_size, tile_width, tile_height, nb, tiles_per_gauss, counts, ccounts);
    SC_LAUNCH_CHECK();
    hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), 0, s, (const unsigned*)counts, nb, isect_offsets,
                       meta_dev);
    SC_LAUNCH_CHECK();
    hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), 0, s, (const unsigned*)ccounts, nb, cstart, cmeta);
    SC_LAUNCH_CHECK();
    hipLaunchKernelGGL(center_scatter_kernel, dim3(grid), dim3(BIN_THREADS), (size_t)nb * 8, s,
                       (const int32_t*)tiles_per_gauss, means2d, radii, CN, N, (float)tile_size, tile_width,
                       tile_height, nb, (const int32_t*)cstart, ccursor, perm);
    SC_LAUNCH_CHECK();
    return SC_OK;
}

extern "C" int sc_isect_bin_sort(const float* means2d, const int32_t* radii, const float* depths, int C,
                                 int N, int tile_size, int tile_width, int tile_height,
                                 const int32_t* isect_offsets, const int64_t* meta_dev,
                                 const void* count_workspace, int64_t capacity, int64_t tile_capacity,
                                 int64_t* isect_ids, int32_t* flatten_ids, void* workspace,
                                 size_t ws_bytes, sc_stream_t stream) {
    if (C < 0 || N < 0 || tile_size <= 0 || tile_width <= 0 || tile_height <= 0 || capacity < 0 ||
        tile_capacity < 0)
        return SC_EINVAL;
    const int64_t CN = (int64_t)C * N;
    const int64_t nb64 = (int64_t)C * tile_width * tile_height;
    if (nb64 > BIN_MAX_BUCKETS || CN > 0x7fffffffLL || capacity > 0x7fffffffLL) return SC_EUNSUPPORTED;
    if (capacity == 0 || CN == 0) return SC_OK;
    const int nb = (int)nb64;
    // LDS of one tile-sort workgroup: 16 B per record must fit ~150 KiB
    if (tile_capacity > 9216) return SC_EUNSUPPORTED;
    int cap = (int)((tile_capacity + 255) / 256 * 256);
    if (cap < 256) cap = 256;
    if (!means2d || !radii || !depths || !isect_offsets || !meta_dev || !count_workspace || !flatten_ids ||
        !workspace)
        return SC_EINVAL;
    if (ws_bytes < sc_isect_bin_workspace_bytes(CN, C, tile_width, tile_height, capacity)) return SC_EWORKSPACE;
    hipStream_t s = sc_s(stream);
    const BinCountLayout L = bin_count_layout(CN, nb);
    const unsigned char* cws = (const unsigned char*)count_workspace;
    const int64_t* cmeta = (const int64_t*)(cws + L.cmeta);
    const int32_t* perm = (const int32_t*)(cws + L.perm);
    unsigned char* ws = (unsigned char*)workspace;
    unsigned* cursor = (unsigned*)ws;
    unsigned char* needs_radix = ws + bin_counts_bytes(nb);
    uint2* bucket = (uint2*)(ws + 2 * bin_counts_bytes(nb));
    SC_HIP(hipMemsetAsync(ws, 0, 2 * bin_counts_bytes(nb), s));   // cursor + needs_radix flags
    const unsigned grid = (unsigned)((CN + BIN_GPB - 1) / BIN_GPB);
    hipLaunchKernelGGL(bin_scatter_kernel, dim3(grid), dim3(BIN_THREADS), (size_t)nb * 8, s, means2d, radii, depths,
                       perm, cmeta, N, (float)tile_size, tile_width, tile_height, nb, isect_offsets, meta_dev,
                       capacity, (int64_t)cap, cursor, bucket, g_sc_debug[0]);
    SC_LAUNCH_CHECK();
    const int tiles_per_cam = tile_width * tile_height;
    const int tile_bits = sc_bits_for(tiles_per_cam);
    const int id_bits = sc_bits_for(CN > 1 ? CN - 1 : 1);
    static bool attr_set = false;
    if (!attr_set) {
        SC_HIP(hipFuncSetAttribute((const void*)tile_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   152 * 1024));
        SC_HIP(hipFuncSetAttribute((const void*)tile_bucket_sort_kernel<0>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        SC_HIP(hipFuncSetAttribute((const void*)tile_bucket_sort_kernel<12>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        attr_set = true;
    }
    // common path: interpolation sort; it flags the (rare) tiles it leaves to the radix kernel
#define SC_LAUNCH_BS(EV)                                                                                      \
    hipLaunchKernelGGL(tile_bucket_sort_kernel<EV>, dim3(nb), dim3(TS_THREADS), (size_t)cap * 16 + 16, s,          \
                       (const uint2*)bucket, isect_offsets, nb, meta_dev, capacity, tiles_per_cam, tile_bits, cap, \
                       needs_radix, isect_ids, flatten_ids, g_sc_debug[2])
    if (cap <= 4 * TS_THREADS) SC_LAUNCH_BS(4);
    else if (cap <= 8 * TS_THREADS) SC_LAUNCH_BS(8);
    else if (cap <= 12 * TS_THREADS) SC_LAUNCH_BS(12);
    else SC_LAUNCH_BS(0);
#undef SC_LAUNCH_BS
    SC_LAUNCH_CHECK();
    hipLaunchKernelGGL(tile_sort_kernel, dim3(nb), dim3(TS_THREADS), (size_t)cap * 16, s, (const uint2*)bucket,
                       isect_offsets, nb, meta_dev, capacity, tiles_per_cam, tile_bits, id_bits, cap,
                       (const unsigned char*)needs_radix, isect_ids, flatten_ids);
    SC_LAUNCH_CHECK();
    return SC_OK;
}
