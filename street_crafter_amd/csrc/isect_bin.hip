// Tile-bucketed intersection path for gfx950: the fast route behind
// gsplat.rendering.isect_tiles(sort=True) + isect_offset_encode
// (street_gaussian/models/street_gaussian_renderer.py:243-253).
//
// The reference-shaped route (isect.hip + radix_sort.hip) moves every (key, value) pair through
// HBM once per radix pass: 6 x 24 B x I.  The sort key is (camera, tile, depth bits) and the
// number of (camera, tile) buckets is small (9600 at 1920x1280), so instead:
//
//   1. bin_count  : every Gaussian adds the 4 corners of its tile rectangle to a 2-D DIFFERENCE
//                   grid (LDS per workgroup, flushed with global atomics); a 2-D prefix sum of
//                   that grid gives the exact number of rectangles covering each tile, and its
//                   exclusive scan is isect_offsets (== isect_offset_encode's lower bounds) -- 4
//                   atomics per Gaussian instead of one per (Gaussian, tile).  The same is done on
//                   the grid of SUPER-TILES (2x2 tiles), plus a histogram of the visible Gaussians
//                   by the super-tile of their rectangle's centre.
//   2. center_scatter : counting sort of the visible Gaussians by that centre super-tile, so that
//                   the next pass walks them in spatial order (stores into a bucket then arrive in
//                   long runs and coalesce; in arrival order they ran ~3x slower).
//   3. bin_scatter: one 8-byte record (depth bits, 4-bit tile mask | flat id) per (Gaussian,
//                   SUPER-tile): 2.9x fewer records than (Gaussian, tile) pairs.  A workgroup
//                   reserves its slice of each bucket with ONE global atomic per (workgroup, bucket).
//   4. super_sort : one workgroup per super-tile sorts its records ONCE on (depth bits, flat id)
//                   in LDS (interpolation sort; radix fallback for degenerate key distributions)
//                   and then emits the up-to-4 per-tile lists by a stable, ballot-compacted filter
//                   on the mask bit -- each emitted list is exactly the stable-sorted list of that
//                   tile.  Stores are coalesced (consecutive lanes -> consecutive slots).
//
// Order contract: (depth bits, flat id) ascending == stable sort of gaussian-major emission order.
// Integer work only: results are bit-identical to the reference-shaped route (tests compare both
// against the oracle).  HBM traffic ~ 16 B x I/2.9 + 12 B x I instead of ~144 B x I.
#include "sc_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int BIN_THREADS = 1024;
constexpr int BIN_GPT = 4;                       // gaussians per thread in the centre pass
constexpr int BIN_GPB = BIN_THREADS * BIN_GPT;
constexpr int CNT_GPT = 8;                       // ... and in the count pass, whose cost is the flush of the per-workgroup
                                                 // grids (A/B on S-1M: 2 -> 36 us, 4 -> 23, 8 -> 19, 16 -> 25).  Small inputs
                                                 // take fewer per thread (template parameter GPT: 1 / 2 / 4): 100 k Gaussians
                                                 // at 8 per thread are 13 workgroups on 256 CUs, and the count pass is in
                                                 // front of the frame's one host wait (14 -> 6 us at S-100k)
constexpr int BIN_MAX_TILES = 36864;             // C * tile_width * tile_height handled by this path (a 3840x2160 frame: 32400)
constexpr int BIN_BIG = 32;                      // rectangles larger than this are walked by a whole wave
constexpr unsigned ID_MASK = 0x0fffffffu;        // flat id lives in the low 28 bits of a record
constexpr unsigned long long KEY_MASK = 0xffffffff0fffffffull;   // (depth, id) without the tile mask

struct Rect { int x0, x1, y0, y1; };

struct Geo {
    int tile_width, tile_height, T;      // tiles
    int ss;                              // super-tile shift (1: 2x2 tiles, 0: super-tile == tile)
    int stw, sth, ST;                    // super-tiles
    int N;
    int ncls;                            // > 0: the PULL route (below) with this many size classes; 0: the scatter route
};

// ---- the PULL route (round 4) -----------------------------------------------------------------------------------
// The scatter route writes one 8-B record per (Gaussian, super-tile) into per-bucket slices of a records buffer
// (bin_scatter_flat: LDS histogram, one global atomic per (workgroup, bucket), scattered 8-B stores) and the sort
// reads them back.  The pull route has no records buffer and no scatter launch: the bucket's own sort workgroup
// GATHERS its records from the spatially sorted payload.  For that the visible Gaussians are counting-sorted by
// (size class, anchor row, anchor column) instead of by centre super-tile: anchor = top-left super-tile of the
// rectangle, class c = smallest c with pull_size(c) >= max(width, height) of the super-tile rectangle
// (half-octave sizes 1 2 3 4 6 8 12 16 24 32 48 64 ...).  A Gaussian of class c can only cover bucket (bx, by) if
// its anchor lies in the s x s window (bx - s, bx] x (by - s, by], s = pull_size(c): per class that is s runs of
// consecutive keys (one per anchor row), i.e. sum of the sizes (~190 for a 60 x 40 grid) contiguous runs of the
// payload per bucket, of which 1 / 1.4 (S-1M, street) .. 1 / 1.7 (a sky set) are hits (tools/an_pull.py).
constexpr int PULL_MAX_CLS = 16;
struct PullTab { int nrows; int row0[PULL_MAX_CLS + 1]; };     // row0[c] = first window row of class c (host-built)

__host__ __device__ __forceinline__ int pull_size(int c) {      // 1 2 3 4 6 8 12 16 24 32 48 64 96 128 192 256
    if (c < 4) return c + 1;
    const int k = c - 4;
    return (k & 1) ? (8 << (k >> 1)) : (6 << (k >> 1));
}
// smallest c with pull_size(c) >= m (m >= 1), clamped to ncls - 1 (the last class's window is the whole grid)
__device__ __forceinline__ int pull_class(int m, int ncls) {
    int c;
    if (m <= 4) c = m - 1;
    else {
        const int p = 31 - __clz(m - 1);                 // floor(log2(m - 1)) >= 2
        c = 4 + 2 * (p - 2) + (((m - 1) >= (6 << (p - 2))) ? 1 : 0);
    }
    return c < ncls - 1 ? c : ncls - 1;
}

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

// the super-tile rectangle covered by a non-empty tile rectangle
__device__ __forceinline__ Rect super_rect(const Rect& r, int ss) {
    Rect s;
    s.x0 = r.x0 >> ss; s.y0 = r.y0 >> ss;
    s.x1 = ((r.x1 - 1) >> ss) + 1; s.y1 = ((r.y1 - 1) >> ss) + 1;
    return s;
}

// ---- view slots (sc_common.h) ----------------------------------------------------------------------------------
// Camera 0's forward axis (third row of the world -> camera rotation) against the registry's slots.  A slot within
// ~7 degrees is this view's (and, with `update`, follows it: a turning camera keeps its slot); otherwise the least
// recently used slot is taken over -- its hints are another view's, valid and only less useful, for one frame.
// Every hint workgroup of a count launch looks the slot up for itself and ONE of them updates the registry: the
// axis is written before the stamp, so a reader sees (old axis, old stamp) -> the same LRU slot, or the new axis
// -> a match on that slot.  Two streams may update at the same time: every word is written whole, a mixed-up
// entry only costs a re-pick.
__device__ __forceinline__ int view_slot_lookup(const float* __restrict__ viewmats, unsigned* registry, bool update) {
    float fx = viewmats[8], fy = viewmats[9], fz = viewmats[10];
    const float inv = rsqrtf(fmaxf(fx * fx + fy * fy + fz * fz, 1e-20f));
    fx *= inv; fy *= inv; fz *= inv;
    const unsigned now = __hip_atomic_load(&registry[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int best = -1, lru = 0;
    float best_dot = 0.9925f;                      // cos(7 degrees)
    unsigned lru_age = 0;
#pragma unroll
    for (int k = 0; k < SC_VIEW_SLOTS; ++k) {
        unsigned e[4];                             // (agent-scope loads: another stream's launch may be updating)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            e[j] = __hip_atomic_load(&registry[4 + 4 * k + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool used = (e[0] | e[1] | e[2]) != 0u;
        const float d = __uint_as_float(e[0]) * fx + __uint_as_float(e[1]) * fy + __uint_as_float(e[2]) * fz;
        if (used && d > best_dot) { best_dot = d; best = k; }
        const unsigned age = used ? now - e[3] + 1u : 0xffffffffu;
        if (age > lru_age) { lru_age = age; lru = k; }
    }
    if (best < 0) best = lru;
    if (update) {
        unsigned* e = registry + 4 + 4 * best;
        atomicExch(&e[0], __float_as_uint(fx) | (fx == 0.f && fy == 0.f && fz == 0.f ? 1u : 0u));
        atomicExch(&e[1], __float_as_uint(fy));
        atomicExch(&e[2], __float_as_uint(fz));
        __threadfence();
        atomicExch(&e[3], atomicAdd(&registry[0], 1u) + 1u);
    }
    return best;
}

// ---- pass 1: counts ---------------------------------------------------------------------------
// dgrid_t : [C][th+1][tw+1]   2-D difference grid over tiles        (-> per-tile counts)
// dgrid_s : [C][sth+1][stw+1] 2-D difference grid over super-tiles  (-> records per super-tile)
// chist   : [C][sth][stw]     visible Gaussians by centre super-tile
// LOCAL: the three grids are accumulated in LDS per workgroup and flushed once (the normal case).  !LOCAL: the
// grids do not fit the LDS (more than ~38 k cells in all, e.g. a 4K frame) and every Gaussian adds straight to
// the global grids (9 global atomics per Gaussian instead of 9 LDS ones + the flush).
template <bool LOCAL, int GPT = CNT_GPT>
__global__ __launch_bounds__(BIN_THREADS) void bin_count_kernel(
    const float* __restrict__ means2d, const int32_t* __restrict__ radii, int64_t CN, Geo g,
    float tile_size, int C, int32_t* __restrict__ tiles_per_gauss, int* __restrict__ dgrid_t,
    int* __restrict__ dgrid_s, unsigned* __restrict__ chist, int n_count_blocks,
    const int32_t* __restrict__ tile_work, unsigned* __restrict__ whint, unsigned* __restrict__ wstat, int blend,
    const float* __restrict__ viewmats, unsigned* view_registry, int32_t* __restrict__ slot_word) {
    extern __shared__ int lds_i[];
    if ((int)blockIdx.x >= n_count_blocks) {
        // HINT blocks (the launch has CUs to spare: ~120 count workgroups): the rasterizer's per-tile work hint for
        // the dispatch list, one thread per tile.  tile_work[t] = the work tile t reported the LAST time a frame of
        // this shape was rasterized; it may be a frame or two old and the camera has moved since, so a tile is filed
        // under the larger of its own value and blend/4 (default 3/4) of the largest value within 2 tiles of it
        // (own value: exact for a camera that stands still; neighbourhood: a heavy region that has moved on).
        // tile_work may be written by another stream's rasterizer right now: any values give a valid list, and the
        // order job (center_scatter_kernel, block 2) reads each hint once.  Also: the maximum and the sum of the raw
        // values.  (This was part of the order job, ONE workgroup that had become the last of its launch to
        // finish: ~10 us of its 21 were these sweeps, instruction-bound on a single CU.)
        const int T = g.T, tw = g.tile_width, th = g.tile_height, n = C * T;
        const int i = ((int)blockIdx.x - n_count_blocks) * BIN_THREADS + (int)threadIdx.x;
        // the hint bank of this call's VIEW (sc_common.h); the slot goes to the rasterizer in the list's last word
        __shared__ int s_view_slot;
        if (threadIdx.x == 0)
            s_view_slot = (viewmats && view_registry)
                              ? view_slot_lookup(viewmats, view_registry, (int)blockIdx.x == n_count_blocks) : 0;
        __syncthreads();
        const int slot = s_view_slot;
        tile_work += (size_t)slot * n;
        if (i == 0) *slot_word = slot;
        unsigned own = 0, w = 0;
        if (i < n) {
            const int cam = i / T, rem = i - cam * T, ty = rem / tw, tx = rem - ty * tw;
            own = (unsigned)min(max(tile_work[i], 0), 65535);
            unsigned m = own;
            if (blend > 0) {
                const int32_t* base = tile_work + cam * T;
                for (int yy = max(ty - 2, 0); yy <= min(ty + 2, th - 1); ++yy)
                    for (int xx = max(tx - 2, 0); xx <= min(tx + 2, tw - 1); ++xx)
                        m = max(m, (unsigned)min(max(base[yy * tw + xx], 0), 65535));
            }
            w = max(own, (m * (unsigned)blend) >> 2);
            whint[i] = w;
        }
        unsigned mx = own, sm = own;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            mx = max(mx, (unsigned)__shfl_xor((int)mx, o, 64));
            sm += (unsigned)__shfl_xor((int)sm, o, 64);
        }
        if (sc_lane() == 0 && mx) { atomicMax(&wstat[0], mx); atomicAdd(&wstat[1], sm); }
        return;
    }
    const int nt = C * (g.tile_height + 1) * (g.tile_width + 1);
    const int ns = g.ss ? C * (g.sth + 1) * (g.stw + 1) : 0;
    // pull route: the histogram is over (camera, size class, anchor) keys, kept as 16-bit halves in LDS (a workgroup
    // has at most 8192 Gaussians)
    const int nc = g.ncls ? (C * g.ncls * g.ST + 1) >> 1 : C * g.ST;
    int* dt = LOCAL ? lds_i : dgrid_t;
    int* ds = LOCAL ? lds_i + nt : dgrid_s;
    int* ch = LOCAL ? lds_i + nt + ns : reinterpret_cast<int*>(chist);
    if (LOCAL) {
        for (int i = threadIdx.x; i < nt + ns + nc; i += BIN_THREADS) lds_i[i] = 0;
        __syncthreads();
    }
    const int64_t base = (int64_t)blockIdx.x * (BIN_THREADS * GPT);
#pragma unroll
    for (int k = 0; k < GPT; ++k) {
        const int64_t i = base + (int64_t)k * BIN_THREADS + threadIdx.x;
        if (i >= CN) continue;
        const float2 m = *reinterpret_cast<const float2*>(means2d + i * 2);
        const Rect r = tile_rect(m.x, m.y, radii[i], tile_size, g.tile_width, g.tile_height);
        const int cnt = (r.y1 - r.y0) * (r.x1 - r.x0);
        tiles_per_gauss[i] = cnt;
        if (cnt <= 0) continue;
        const int cam = (int)(i / g.N);
        {
            const int w = g.tile_width + 1;
            int* d = dt + cam * (g.tile_height + 1) * w;
            atomicAdd(&d[r.y0 * w + r.x0], 1);  atomicAdd(&d[r.y0 * w + r.x1], -1);
            atomicAdd(&d[r.y1 * w + r.x0], -1); atomicAdd(&d[r.y1 * w + r.x1], 1);
        }
        const Rect s = super_rect(r, g.ss);
        if (g.ss) {
            const int w = g.stw + 1;
            int* d = ds + cam * (g.sth + 1) * w;
            atomicAdd(&d[s.y0 * w + s.x0], 1);  atomicAdd(&d[s.y0 * w + s.x1], -1);
            atomicAdd(&d[s.y1 * w + s.x0], -1); atomicAdd(&d[s.y1 * w + s.x1], 1);
        }
        if (g.ncls) {
            const int key = ((cam * g.ncls + pull_class(max(s.x1 - s.x0, s.y1 - s.y0), g.ncls)) * g.sth + s.y0) * g.stw + s.x0;
            if (LOCAL) atomicAdd(&ch[key >> 1], 1 << ((key & 1) << 4));
            else atomicAdd(&chist[key], 1u);
        } else {
            const int cx = (s.x0 + s.x1 - 1) >> 1, cy = (s.y0 + s.y1 - 1) >> 1;
            atomicAdd(&ch[cam * g.ST + cy * g.stw + cx], 1);
        }
    }
    if (!LOCAL) return;
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += BIN_THREADS) { const int v = dt[i]; if (v) atomicAdd(&dgrid_t[i], v); }
    for (int i = threadIdx.x; i < ns; i += BIN_THREADS) { const int v = ds[i]; if (v) atomicAdd(&dgrid_s[i], v); }
    if (g.ncls) {
        for (int i = threadIdx.x; i < nc; i += BIN_THREADS) {
            const unsigned v = (unsigned)ch[i];
            if (v & 0xffffu) atomicAdd(&chist[2 * i], v & 0xffffu);
            if (v >> 16) atomicAdd(&chist[2 * i + 1], v >> 16);
        }
    } else {
        for (int i = threadIdx.x; i < nc; i += BIN_THREADS) { const int v = ch[i]; if (v) atomicAdd(&chist[i], (unsigned)v); }
    }
}

// block-wide exclusive scan helper (1024 threads): returns the exclusive prefix of `sum`, the
// grand total in *total and the block maximum of `mx` in *maxv.
__device__ __forceinline__ long long block_scan_1024(long long sum, unsigned mx, long long* total,
                                                     unsigned* maxv, long long* wave_tot, unsigned* wave_max) {
    const int lane = sc_lane(), wave = threadIdx.x >> 6;
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
    *total = tot;
    *maxv = m;
    return run;
}

// In-LDS inclusive prefix along one strided line of n cells; 8 consecutive lanes share a line
// (each scans a segment serially, segment totals are combined with width-8 shuffles).  Every lane
// of the wave must call it; lanes with active == false touch no memory.
__device__ __forceinline__ void line_prefix8(int* base, int n, int stride, int sub, bool active) {
    const int seg = (n + 7) >> 3;
    const int b = sub * seg, e = min(n, b + seg);
    int run = 0;
    if (active)
        for (int i = b; i < e; ++i) { run += base[i * stride]; base[i * stride] = run; }
    int incl = run;
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
        const int tmp = __shfl_up(incl, d, 8);
        if (sub >= d) incl += tmp;
    }
    const int off = incl - run;
    if (active && off)
        for (int i = b; i < e; ++i) base[i * stride] += off;
}

// Single-workgroup scan jobs (run as extra blocks of the centre-scatter launch):
//   0: tile difference grid  -> per-tile counts -> exclusive scan = isect_offsets; meta[0..1]
//   1: super-tile difference grid -> record offsets;                               meta[2..3]
// (is_grid == 0: plain exclusive scan of a count array)
// A difference grid [C][gh+1][gw+1] becomes counts by a 2-D prefix sum (x then y).
struct ScanJob { const int* src; int C, gw, gh; int32_t* out; int64_t* meta; int is_grid; };
struct ScanJobs { ScanJob j[3]; };

__device__ __forceinline__ void run_scan_job(const ScanJob& job, int* grid, long long* wave_tot,
                                             unsigned* wave_max) {
    const int t = threadIdx.x;
    const int gw = job.gw, gh = job.gh, C = job.C;
    const int n = C * gh * gw;
    const int per = (n + 1023) / 1024;
    const int beg = min(t * per, n), end = min(beg + per, n);
    long long sum = 0;
    unsigned mx = 0;
    long long tot;
    unsigned m;
    if (job.is_grid) {
        const int W = gw + 1, H = gh + 1, ncell = C * H * W;
        for (int i = t; i < ncell; i += 1024) grid[i] = job.src[i];
        __syncthreads();
        const int sub = t & 7;
        for (int l0 = 0; l0 < C * H; l0 += 128) {           // prefix along x (rows)
            const int l = l0 + (t >> 3);
            line_prefix8(grid + min(l, C * H - 1) * W, W, 1, sub, l < C * H);
        }
        __syncthreads();
        for (int l0 = 0; l0 < C * W; l0 += 128) {           // prefix along y (columns)
            const int l = min(l0 + (t >> 3), C * W - 1);
            const int cam = l / W, x = l - cam * W;
            line_prefix8(grid + cam * H * W + x, H, W, sub, l0 + (t >> 3) < C * W);
        }
        __syncthreads();
        // exclusive scan over (camera, row, column); the cell index walks incrementally
        int cam = beg / (gh * gw), rem = beg - cam * gh * gw;
        int y = rem / gw, x = rem - y * gw;
        const int cam0 = cam, y0 = y, x0 = x;
        for (int i = beg; i < end; ++i) {
            const unsigned c = (unsigned)grid[(cam * H + y) * W + x];
            sum += c;
            mx = max(mx, c);
            if (++x == gw) { x = 0; if (++y == gh) { y = 0; ++cam; } }
        }
        long long run = block_scan_1024(sum, mx, &tot, &m, wave_tot, wave_max);
        cam = cam0; y = y0; x = x0;
        for (int i = beg; i < end; ++i) {
            job.out[i] = (int32_t)run;
            run += (unsigned)grid[(cam * H + y) * W + x];
            if (++x == gw) { x = 0; if (++y == gh) { y = 0; ++cam; } }
        }
    } else {
        const unsigned* counts = reinterpret_cast<const unsigned*>(job.src);
        for (int i = beg; i < end; ++i) { const unsigned c = counts[i]; sum += c; mx = max(mx, c); }
        long long run = block_scan_1024(sum, mx, &tot, &m, wave_tot, wave_max);
        for (int i = beg; i < end; ++i) { job.out[i] = (int32_t)run; run += counts[i]; }
    }
    if (t == 0) { job.meta[0] = tot; job.meta[1] = (long long)m; }
}


// ---- spatial order: counting sort of the visible Gaussians by centre super-tile --------------------
// The host needs meta (the output sizes) once per frame.  Instead of a D2H copy + event on the stream
// (a copy kernel plus a ~6 us barrier bubble between the count and the scatter kernels), one lane
// stores the four numbers straight into host-mapped pinned memory, then a sequence number with
// system-scope release; the host polls the sequence number.
__device__ __forceinline__ void publish_meta(const int64_t* __restrict__ meta_dev, int64_t* mirror, int64_t seq) {
    mirror[0] = meta_dev[0]; mirror[1] = meta_dev[1]; mirror[2] = meta_dev[2]; mirror[3] = meta_dev[3];
    __hip_atomic_store(&mirror[4], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the plain dispatch list (every tile whole, in tile order; no second list) for a frame without Gaussians
__global__ void plain_order_kernel(int32_t* out, int n_tiles, int n_total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total) return;
    // n_total - 2: "no second list"; n_total - 1: the view slot (nothing is rasterized: slot 0)
    out[i] = i < n_tiles ? i << 2 : (i >= n_total - 2 ? 0 : -1);
}

__global__ void publish_meta_kernel(const int64_t* __restrict__ meta_dev, int64_t* mirror, int64_t seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) publish_meta(meta_dev, mirror, seq);
}

// The launch also carries the two grid scans (blocks 0 and 1): they depend on the count
// pass only, like this kernel, so they run beside it instead of in front of it (the separate scan
// launch sat 13 us on the critical path).  Every centre workgroup builds the bucket starts itself, by a
// block scan of the 2400-entry centre histogram.  The scan block that finishes second publishes meta.
__global__ __launch_bounds__(BIN_THREADS) void center_scatter_kernel(
    const int32_t* __restrict__ tiles_per_gauss, const float* __restrict__ means2d,
    const int32_t* __restrict__ radii, int64_t CN, Geo g, float tile_size, int n_sbuckets,
    const float* __restrict__ depths, const unsigned* __restrict__ chist, unsigned* __restrict__ ccursor,
    uint4* __restrict__ sorted, int64_t* __restrict__ cmeta, int32_t* __restrict__ cstart, ScanJobs jobs,
    unsigned* __restrict__ scans_done, int64_t* __restrict__ meta_dev, int64_t* meta_mirror, int64_t seq,
    const unsigned* __restrict__ whint, const unsigned* __restrict__ wstat, int32_t* __restrict__ tile_order,
    int n_tiles_total, int staged, int split_pct, int want_bwd SC_DIAG_PARAM(odbg)) {
    extern __shared__ unsigned lds[];
    __shared__ long long wave_tot[16];
    __shared__ unsigned wave_max[16];
    // the scan blocks come first in dispatch order, so that meta is published early even when the
    // centre workgroups need several rounds
    if (blockIdx.x < 2) {
        run_scan_job(jobs.j[blockIdx.x], reinterpret_cast<int*>(lds), wave_tot, wave_max);
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicAdd(scans_done, 1u) == 1u && meta_mirror) {       // both halves of meta are in place
                __threadfence();
                publish_meta(meta_dev, meta_mirror, seq);
            }
        }
        return;
    }
    if (blockIdx.x == 2) {
        // The rasterizer's DISPATCH LIST (include/street_crafter_amd.h, sc_rasterize_fwd): longest-running tiles
        // first, and the tiles whose walk would be the launch's tail shared by two waves (16 x 8 halves).
        // tile_work[t] = the work (blend iterations + 8 per staged batch) tile t reported the LAST time a frame of
        // this shape was rasterized -- a scheduling hint: the values may be a frame or two old or half updated; any
        // values give a valid list.  The launch's makespan is one tile's serial walk plus the throughput part:
        // heaviest-first took S-1M 150 -> 140 us and the street scene 292 -> 234 us, halving every tile within a
        // factor 2 of the heaviest (at most one tile in eight) a further -> 133 / 152 us, with exact hints
        // (tools/exp_raster_split.py).  Counting sort over 1024 classes of work, heaviest class first; the order
        // inside a class is whatever the atomics give (tile order and random order time the same).
        if (!tile_order) return;
        // This ONE workgroup must stay shorter than the centre workgroups beside it (~20 us): the per-tile hints come
        // ready-made from the count launch's hint blocks (whint, wstat), LDS atomics are aggregated over runs of
        // equal classes, the list is put together in LDS and written out coalesced.
        unsigned* cls = lds;                 // [1024]
        const int tpad = (n_tiles_total + 1) & ~1;
        unsigned short* snap = reinterpret_cast<unsigned short*>(lds + 1024);     // [tiles] class of every tile
        __shared__ int s_lim, s_nsplit, s_tail, s_before;
        const int n_items_max = n_tiles_total + n_tiles_total / 8 + 8;             // == sc_tile_order_fwd_items
        const int cap = n_tiles_total / 8;
        const int lane = sc_lane();
        cls[threadIdx.x] = 0;
        if (threadIdx.x == 0) { s_lim = 0; s_nsplit = 0; s_tail = 1024; s_before = 0x7fffffff; }
#ifdef SC_DIAG      // stage stamps of this one workgroup (tools/exp_order_job.py), diagnostic build only
        __shared__ long long s_stamp[10];
        auto stamp = [&](int k) { if ((odbg & 64) && threadIdx.x == 0) s_stamp[k] = (long long)wall_clock64(); };
#else
        auto stamp = [](int) {};
#endif
        stamp(0);
        const unsigned wmax = whint ? wstat[0] : 0u;
        const long long wsum = whint ? (long long)wstat[1] : 0;
        // halving pays when a few tiles stand far above the rest (street scene: heaviest 7x the mean, -70 us); on
        // an even frame (S-1M: 1.7x) it only adds the halves' second staging (+3 us): ask for 3x the mean
        const bool skewed = (long long)wmax * n_tiles_total >= 3 * wsum;
        long long tot;
        stamp(1);
        if (SC_DIAG_BIT(odbg, 1)) return;          // diagnostic (debug1 bit 16): price the start
        int shift = 0;
        while ((wmax >> shift) > 1023u) ++shift;
        stamp(2);
        if (SC_DIAG_BIT(odbg, 2)) return;
        // (neighbouring tiles = neighbouring lanes often share a class, and 64 lanes adding to one LDS counter
        // serialise.  A RUN of equal classes among consecutive lanes is added by its first lane alone, in both sweeps.)
        auto run_of = [&](int c, int* head_lane, int* len) {
            const int prev = __shfl_up(c, 1, 64);
            const unsigned long long heads = __ballot(lane == 0 || c != prev);
            const unsigned long long rest = lane == 63 ? 0ull : heads >> (lane + 1);
            *len = rest ? __ffsll((long long)rest) : 64 - lane;               // meaningful on head lanes
            *head_lane = 63 - __clzll((long long)(heads & (sc_lanemask_lt() | (1ull << lane))));
        };
        const int n_round = (n_tiles_total + 63) & ~63;                       // whole waves take part in the votes
#pragma unroll 2
        for (int i = threadIdx.x; i < n_round; i += BIN_THREADS) {
            int c = -1;
            if (i < n_tiles_total) {
                const unsigned w = whint ? min(whint[i], 65535u) : 0u;        // read ONCE: it decides the tile's place
                c = 1023 - (int)min(1023u, w >> shift);
                snap[i] = (unsigned short)c;
            }
            int hl, len;
            run_of(c, &hl, &len);
            if (hl == lane && c >= 0) atomicAdd(&cls[c], (unsigned)len);
        }
        __syncthreads();
        stamp(3);
        if (SC_DIAG_BIT(odbg, 4)) return;          // ... + the class sweep
        // the forward list is staged in LDS behind the class table (2 B per item) when the launch provides the room
        unsigned short* stage = staged ? snap + tpad : nullptr;
        const unsigned cnt = cls[threadIdx.x];
        unsigned mx;
        const long long run = block_scan_1024((long long)cnt, 0u, &tot, &mx, wave_tot, wave_max);
        // Which tiles are listed as two halves (at most `cap` of them, whole classes only):
        //  * a SKEWED frame: classes [0, lim), i.e. the tiles whose work is at least split_pct % of the heaviest
        //    tile's, heaviest classes first -- their walk would be the launch's tail;
        //  * an even frame: classes [tail_c, 1024), i.e. the LAST tiles of the list -- smaller items at the end of
        //    the launch even out the ragged last round (S-1M 137.5 -> 132.4 us, whatever the hint is worth).
        const int c_split = 1023 - (int)((((unsigned long long)wmax * (unsigned)split_pct) / 100u) >> shift);
        const bool on = split_pct > 0 && wmax >= 32u;
        const bool mine = on && skewed && (int)threadIdx.x <= c_split && run + cnt <= cap;
        const bool mine_tail = on && !skewed && (long long)n_tiles_total - run <= cap;
        // `mine` holds on a prefix of the classes, `mine_tail` on a suffix (run is monotone): the boundary threads
        // report, nobody else touches the shared words (a thousand same-address LDS atomics cost microseconds)
        {
            const unsigned long long bm = __ballot(mine), bt = __ballot(mine_tail);
            const bool next_mine = lane < 63 ? ((bm >> (lane + 1)) & 1ull) != 0 : false;      // lane 63: decided below
            const bool last_here = mine && !next_mine;             // candidate for "last class of the prefix"
            const bool prev_tail = lane > 0 ? ((bt >> (lane - 1)) & 1ull) != 0 : false;
            const bool first_here = mine_tail && !prev_tail;       // candidate for "first class of the suffix"
            // a candidate at a wave boundary may continue in the neighbouring wave: max / min over the candidates decides
            if (last_here) { atomicMax(&s_lim, (int)threadIdx.x + 1); atomicMax(&s_nsplit, (int)(run + cnt)); }
            if (first_here) { atomicMin(&s_tail, (int)threadIdx.x); atomicMin(&s_before, (int)run); }
        }
        __syncthreads();
        const int lim = s_lim;                            // `mine` holds on a prefix of the classes,
        const int tail_c = s_tail, n_before = min(s_before, n_tiles_total);     // `mine_tail` on a suffix (run is monotone)
        const int n_split = tail_c < 1024 ? n_tiles_total - n_before : s_nsplit;
        {
            const int c = (int)threadIdx.x;
            unsigned base;
            if (c < lim) base = (unsigned)(2 * run);
            else if (c >= tail_c) base = (unsigned)(n_before + 2 * (run - n_before));
            else base = (unsigned)(run + (tail_c < 1024 ? 0 : n_split));
            cls[c] = base;
        }
        __syncthreads();
        stamp(4);
        if (SC_DIAG_BIT(odbg, 8)) return;          // ... + the class scan and the split decision
        for (int i = threadIdx.x; i < n_round; i += BIN_THREADS) {
            const int c = i < n_tiles_total ? (int)snap[i] : -1;
            int hl, len;
            run_of(c, &hl, &len);
            const unsigned parts = (c < lim || c >= tail_c) ? 2u : 1u;
            unsigned slot = 0;
            if (hl == lane && c >= 0) slot = atomicAdd(&cls[c], (unsigned)len * parts);
            slot = (unsigned)__shfl((int)slot, hl, 64) + (unsigned)(lane - hl) * parts;
            if (c < 0) continue;
            if (stage) {                   // the list is put together in LDS and written out in one coalesced sweep
                if (slot + parts <= (unsigned)n_items_max) {
                    stage[slot] = (unsigned short)i;
                    if (parts == 2u) stage[slot + 1] = (unsigned short)i;
                }
            } else if (parts == 2u) {
                if (slot + 1 < (unsigned)n_items_max) { tile_order[slot] = i << 2 | 1; tile_order[slot + 1] = i << 2 | 2; }
            } else {
                if (slot < (unsigned)n_items_max) tile_order[slot] = i << 2;
            }
            // the BACKWARD's list (whole tiles only, same order) follows the forward's: a kernel that skipped every
            // second item of a run of halves would leave half the XCDs without work there (blocks go round-robin
            // over the XCDs): 284 -> 299 us on the training step's backward
            // (built only when the backward is set to take it, raster_bwd_split 0: this workgroup's scattered
            // 4-B stores are what it spends most of its time on, and it is the last to finish on small inputs)
            if (want_bwd) {
                unsigned bslot;
                if (c < lim) bslot = slot >> 1;
                else if (c >= tail_c) bslot = (unsigned)n_before + ((slot - (unsigned)n_before) >> 1);
                else bslot = slot - (tail_c < 1024 ? 0u : (unsigned)n_split);
                if (bslot < (unsigned)n_tiles_total) tile_order[n_items_max + bslot] = i << 2;
            }
        }
        if (threadIdx.x == 0) tile_order[n_items_max + n_tiles_total] = want_bwd;       // "the second list is there"
        const int n_items = n_tiles_total + n_split;
        stamp(5);
        if (SC_DIAG_BIT(odbg, 16)) return;         // ... + the scatter sweep
        if (stage) {
            // (this ONE workgroup's scattered 4-B global stores were what it spent most of its time on -- it had become
            // the last of the launch to finish: center_scatter 24 -> 29 us on S-1M.)  A tile listed as halves sits in
            // two neighbouring slots, so the kind of an item follows from its neighbours.
            __syncthreads();
            for (int j = threadIdx.x; j < n_items_max; j += BIN_THREADS) {
                int item = -1;
                if (j < n_items) {
                    const int tl = stage[j];
                    const int kind = (j + 1 < n_items && stage[j + 1] == tl) ? 1 : ((j > 0 && stage[j - 1] == tl) ? 2 : 0);
                    item = tl << 2 | kind;
                }
                tile_order[j] = item;
            }
        } else {
            for (int i = n_items + threadIdx.x; i < n_items_max; i += BIN_THREADS) tile_order[i] = -1;
        }
        stamp(6);
#ifdef SC_DIAG
        if ((odbg & 64) && threadIdx.x == 0) {          // diagnostic: stage times in 10-ns ticks at the end of the buffer's forward part
            for (int k = 0; k < 7; ++k) tile_order[n_items_max - 8 + k] = (int)(s_stamp[k] - s_stamp[0]);
        }
#endif
        return;
    }
    const int cblock = (int)blockIdx.x - 3;
    if (g.ncls) {
        // PULL route: counting sort by (camera, size class, anchor row, anchor column).  Every workgroup scans the key
        // histogram itself (as the scatter route's workgroups scan the centre histogram: a scan of its own would be one more
        // dependent launch); a Gaussian's slot is its key's start plus a global running cursor (in arrival order nearly every
        // Gaussian of a workgroup has a key of its own, so a per-workgroup LDS aggregation would save nothing).
        const int nkeys = n_sbuckets * g.ncls;
        unsigned* cst = lds;                 // [nkeys]
        for (int i = threadIdx.x; i < nkeys; i += BIN_THREADS) cst[i] = chist[i];
        __syncthreads();
        const int per = ((nkeys + BIN_THREADS - 1) / BIN_THREADS) | 1;      // odd: the threads' strides hit distinct banks
        const int beg = min((int)threadIdx.x * per, nkeys), end = min(beg + per, nkeys);
        long long sum = 0;
        for (int i = beg; i < end; ++i) sum += cst[i];
        long long tot;
        unsigned mx;
        long long run = block_scan_1024(sum, 0u, &tot, &mx, wave_tot, wave_max);
        for (int i = beg; i < end; ++i) { const unsigned c = cst[i]; cst[i] = (unsigned)run; run += c; }
        if (cblock == 0 && threadIdx.x == 0) { cmeta[0] = tot; cmeta[1] = 0; }   // number of visible Gaussians
        __syncthreads();
        if (cblock == 0) {                   // the key starts, for the pulling workgroups of the sort launch
            for (int i = threadIdx.x; i < nkeys; i += BIN_THREADS) cstart[i] = (int32_t)cst[i];
            if (threadIdx.x == 0) cstart[nkeys] = (int32_t)tot;
        }
        const int64_t base = (int64_t)cblock * BIN_GPB;
        int key[BIN_GPT];
        uint4 pay[BIN_GPT];
#pragma unroll
        for (int k = 0; k < BIN_GPT; ++k) {
            const int64_t i = base + (int64_t)k * BIN_THREADS + threadIdx.x;
            key[k] = -1;
            pay[k] = make_uint4(0u, 0u, 0u, 0u);
            if (i < CN && tiles_per_gauss[i] > 0) {
                const float2 m = *reinterpret_cast<const float2*>(means2d + i * 2);
                const Rect r = tile_rect(m.x, m.y, radii[i], tile_size, g.tile_width, g.tile_height);
                const Rect s = super_rect(r, g.ss);
                key[k] = (((int)(i / g.N) * g.ncls + pull_class(max(s.x1 - s.x0, s.y1 - s.y0), g.ncls)) * g.sth + s.y0) * g.stw + s.x0;
                pay[k] = make_uint4((unsigned)r.x0 | ((unsigned)r.x1 << 16), (unsigned)r.y0 | ((unsigned)r.y1 << 16),
                                    __float_as_uint(depths[i]), (unsigned)i);
            }
        }
        unsigned slot[BIN_GPT];
#pragma unroll
        for (int k = 0; k < BIN_GPT; ++k) slot[k] = key[k] >= 0 ? cst[key[k]] + atomicAdd(&ccursor[key[k]], 1u) : 0u;
#pragma unroll
        for (int k = 0; k < BIN_GPT; ++k)
            if (key[k] >= 0 && slot[k] < (unsigned)CN) sorted[slot[k]] = pay[k];
        return;
    }
    unsigned* hist = lds;                    // [n_sbuckets]
    unsigned* gbase = lds + n_sbuckets;      // [n_sbuckets]
    unsigned* cst = lds + 2 * n_sbuckets;    // [n_sbuckets] first slot of each centre bucket in the spatial order
    {
        const int per = (n_sbuckets + BIN_THREADS - 1) / BIN_THREADS;
        const int beg = min((int)threadIdx.x * per, n_sbuckets), end = min(beg + per, n_sbuckets);
        long long sum = 0;
        for (int i = beg; i < end; ++i) sum += chist[i];
        long long tot;
        unsigned mx;
        long long run = block_scan_1024(sum, 0u, &tot, &mx, wave_tot, wave_max);
        for (int i = beg; i < end; ++i) { cst[i] = (unsigned)run; run += chist[i]; }
        if (cblock == 0 && threadIdx.x == 0) { cmeta[0] = tot; cmeta[1] = 0; }   // number of visible Gaussians
    }
    for (int b = threadIdx.x; b < n_sbuckets; b += BIN_THREADS) hist[b] = 0;
    __syncthreads();
    const int64_t base = (int64_t)cblock * BIN_GPB;
    int cb[BIN_GPT];
    uint4 pay[BIN_GPT];      // tile rectangle (x0 | x1 << 16, y0 | y1 << 16), depth bits, flat id
#pragma unroll
    for (int k = 0; k < BIN_GPT; ++k) {
        const int64_t i = base + (int64_t)k * BIN_THREADS + threadIdx.x;
        cb[k] = -1;
        pay[k] = make_uint4(0u, 0u, 0u, 0u);
        if (i < CN && tiles_per_gauss[i] > 0) {
            const float2 m = *reinterpret_cast<const float2*>(means2d + i * 2);
            const Rect r = tile_rect(m.x, m.y, radii[i], tile_size, g.tile_width, g.tile_height);
            const Rect s = super_rect(r, g.ss);
            const int cx = (s.x0 + s.x1 - 1) >> 1, cy = (s.y0 + s.y1 - 1) >> 1;
            cb[k] = (int)(i / g.N) * g.ST + cy * g.stw + cx;
            atomicAdd(&hist[cb[k]], 1u);
            pay[k] = make_uint4((unsigned)r.x0 | ((unsigned)r.x1 << 16), (unsigned)r.y0 | ((unsigned)r.y1 << 16),
                                __float_as_uint(depths[i]), (unsigned)i);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < n_sbuckets; b += BIN_THREADS) {
        const unsigned c = hist[b];
        if (c) gbase[b] = cst[b] + atomicAdd(&ccursor[b], c);
        hist[b] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BIN_GPT; ++k)
        if (cb[k] >= 0) sorted[gbase[cb[k]] + atomicAdd(&hist[cb[k]], 1u)] = pay[k];
}

// ---- pass 2: records ---------------------------------------------------------------------------------
// One 8-B record (depth bits | tile mask << 28 + id) per (Gaussian, super-tile).  Input: the spatially
// ordered payload center_scatter wrote (16 B per visible Gaussian, read coalesced: gathering means /
// radii / depths through a permutation cost 30 us of random 64-B sectors).  The (Gaussian, super-tile)
// pairs of a wave's 64 Gaussians are FLATTENED: an exclusive scan of the rectangle sizes numbers the
// pairs 0..T-1, a byte table in LDS maps pair -> owning lane, and lane l handles pairs l, l + 64, ...
// so every step has 64 busy lanes (the average rectangle has ~9 super-tiles, the largest of a wave up
// to BIN_BIG).  Rectangles above BIN_BIG are walked by the whole wave, 64 super-tiles per step.
// Slots: per-workgroup LDS histogram, one global atomic per (workgroup, touched bucket), LDS cursors.
constexpr int FLAT_THREADS = 512;                // A/B on S-1M: 256 -> 52 us, 512 -> 47, 1024 -> 51
constexpr int FLAT_THREADS_SMALL = 128;          // <= 262144 Gaussians: 2 waves x 16 Gaussians per workgroup
struct FlatTab {                 // per wave; one 16-B LDS read per table and pair (they were 12 separate arrays)
    uint4 a[64];     // super-tile rectangle: x0 | y0 << 16, (65536 / width + 1) | width << 20, first pair number, camera bucket base
    uint4 b[64];     // the Gaussian's payload as center_scatter wrote it: tile x0 | x1 << 16, y0 | y1 << 16, depth bits, id
    unsigned char owner[64 * BIN_BIG];             // pair number -> lane
};

// GPW = Gaussians per wave (64, or 16 for small inputs: a wave walks its big rectangles one after the other, and
// a set of a few ten thousand huge splats -- a sky -- otherwise leaves most SIMDs without a wave)
template <int FLAT_THREADS, int GPW>
__global__ __launch_bounds__(FLAT_THREADS) void bin_scatter_flat_kernel(
    const uint4* __restrict__ sorted, const int64_t* __restrict__ n_visible, Geo g, int n_sbuckets,
    const int32_t* __restrict__ soffsets, const int64_t* __restrict__ meta, int64_t capacity,
    int64_t rec_capacity, int64_t super_capacity, unsigned* __restrict__ cursor,
    uint2* __restrict__ records SC_DIAG_PARAM(dbg)) {
    extern __shared__ unsigned lds[];
    // the caller may have sized the buffers from a prediction: do nothing if they are too small
    if (meta[0] > capacity || meta[2] > rec_capacity || meta[3] > super_capacity) return;
    const int64_t M = n_visible[0];
    constexpr int GPB = FLAT_THREADS / 64 * GPW;           // Gaussians per workgroup
    const int64_t base_j = (int64_t)blockIdx.x * GPB;
    if (base_j >= M) return;
    unsigned* hist = lds;                  // [n_sbuckets] counts, then running local cursors
    unsigned* gbase = lds + n_sbuckets;    // [n_sbuckets] global start of this workgroup's slice
    const int lane = sc_lane(), wave = threadIdx.x >> 6;
    FlatTab& tab = reinterpret_cast<FlatTab*>(lds + ((2 * n_sbuckets + 3) & ~3))[wave];      // 16-B aligned
    for (int b = threadIdx.x; b < n_sbuckets; b += FLAT_THREADS) hist[b] = 0;

    // consecutive lanes take consecutive Gaussians of the spatial order
    const int64_t j = base_j + (int64_t)wave * GPW + lane;
    Rect r = {0, 0, 0, 0}, sr = {0, 0, 0, 0};
    int cam_base = 0;
    unsigned depth = 0, id = 0;
    uint4 pay = make_uint4(0u, 0u, 0u, 0u);
    if (lane < GPW && j < M) {
        pay = sorted[j];
        r.x0 = (int)(pay.x & 0xffffu); r.x1 = (int)(pay.x >> 16);
        r.y0 = (int)(pay.y & 0xffffu); r.y1 = (int)(pay.y >> 16);
        depth = pay.z;
        id = pay.w;
        cam_base = (int)(id / (unsigned)g.N) * g.ST;
        if ((r.x1 - r.x0) * (r.y1 - r.y0) > 0) sr = super_rect(r, g.ss);
    }
    const int sw = sr.x1 - sr.x0, cnt = sw * (sr.y1 - sr.y0);
    const bool live = cnt > 0;
    const int c = (live && cnt <= BIN_BIG) ? cnt : 0;
    const int incl = sc_wave_incl_scan(c);
    const int off = incl - c;
    const int T = __builtin_amdgcn_readlane(incl, 63);
    // (only rectangles of at most BIN_BIG super-tiles are decoded through the table: width <= BIN_BIG;
    // q / width == (q * inv) >> 16 for q < 1024)
    tab.a[lane] = make_uint4((unsigned)sr.x0 | (unsigned)sr.y0 << 16,
                             c ? ((unsigned)(65536 / sw + 1) | (unsigned)sw << 20) : 0u, (unsigned)off, (unsigned)cam_base);
    tab.b[lane] = pay;
    for (int q = 0; q < c; ++q) tab.owner[off + q] = (unsigned char)lane;
    __syncthreads();                       // hist zeroed, tables complete
    if (SC_DIAG_BIT(dbg, 4)) return;       // diagnostic: price the load + table build alone

    // decode pair number p -> bucket (and, for pass 2, everything the record needs)
    auto bucket_of = [&](int p, int& o, int& sx, int& sy) -> int {
        o = tab.owner[p];
        const uint4 A = tab.a[o];
        const int q = p - (int)A.z;
        const int wdt = (int)(A.y >> 20);
        const int row = (q * (int)(A.y & 0xfffffu)) >> 16;
        sy = (int)(A.x >> 16) + row;
        sx = (int)(A.x & 0xffffu) + (q - row * wdt);
        return (int)A.w + sy * g.stw + sx;
    };
    // large rectangles: the wave walks them together, 64 super-tiles per step
    auto walk_big = [&](auto&& f) {
        unsigned long long big = __ballot(cnt > BIN_BIG);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            Rect br;
            br.x0 = __builtin_amdgcn_readlane(r.x0, src); br.x1 = __builtin_amdgcn_readlane(r.x1, src);
            br.y0 = __builtin_amdgcn_readlane(r.y0, src); br.y1 = __builtin_amdgcn_readlane(r.y1, src);
            const int bx0 = __builtin_amdgcn_readlane(sr.x0, src), by0 = __builtin_amdgcn_readlane(sr.y0, src);
            const int bw = __builtin_amdgcn_readlane(sw, src), bcnt = __builtin_amdgcn_readlane(cnt, src);
            const int bbase = __builtin_amdgcn_readlane(cam_base, src);
            const unsigned bd = (unsigned)__builtin_amdgcn_readlane((int)depth, src);
            const unsigned bi = (unsigned)__builtin_amdgcn_readlane((int)id, src);
            // q / bw by multiplication: exact for q < 2^32 / bw (an integer division costs ~40 instructions, and a
            // set of big splats -- a sky -- is walked entirely through this loop)
            // (bw == 1 -- a big splat clipped to ONE column of super-tiles at the left or right edge of the frame --
            // has no 32-bit reciprocal: 2^32 / 1 wraps to 0 and every cell landed in row 0, i.e. in the wrong buckets)
            const unsigned binv = bw > 1 ? 0xffffffffu / (unsigned)bw + 1u : 0u;
            for (int q = lane; q < bcnt; q += 64) {
                const int row = bw > 1 ? (int)__umulhi((unsigned)q, binv) : q;
                const int sy = by0 + row, sx = bx0 + (q - row * bw);
                f(bbase + sy * g.stw + sx, br, sx, sy, bd, bi);
            }
        }
    };

    // pass 1: counts
    for (int p = lane; p < T; p += 64) {
        int o, sx, sy;
        atomicAdd(&hist[bucket_of(p, o, sx, sy)], 1u);
    }
    walk_big([&](int b, const Rect&, int, int, unsigned, unsigned) { atomicAdd(&hist[b], 1u); });
    __syncthreads();
    if (SC_DIAG_BIT(dbg, 8)) return;       // diagnostic: ... and the counting pass
    for (int b = threadIdx.x; b < n_sbuckets; b += FLAT_THREADS) {
        const unsigned cb = hist[b];
        if (cb) gbase[b] = (unsigned)soffsets[b] + atomicAdd(&cursor[b], cb);
        hist[b] = 0;
    }
    __syncthreads();
    if (SC_DIAG_BIT(dbg, 2)) return;
    // pass 2: slots + records
    auto tile_mask = [&](int x0, int x1, int y0, int y1, int sx, int sy) -> unsigned {
        if (!g.ss) return 1u;
        unsigned m = 0;
        const int tx = sx << 1, ty = sy << 1;
        const bool cx0 = tx >= x0 && tx < x1, cx1 = tx + 1 >= x0 && tx + 1 < x1;
        if (ty >= y0 && ty < y1) { if (cx0) m |= 1u; if (cx1) m |= 2u; }
        if (ty + 1 >= y0 && ty + 1 < y1) { if (cx0) m |= 4u; if (cx1) m |= 8u; }
        return m;
    };
    for (int p = lane; p < T; p += 64) {
        int o, sx, sy;
        const int b = bucket_of(p, o, sx, sy);
        const unsigned slot = gbase[b] + atomicAdd(&hist[b], 1u);
        const uint4 B = tab.b[o];
        const unsigned mask = tile_mask((int)(B.x & 0xffffu), (int)(B.x >> 16), (int)(B.y & 0xffffu), (int)(B.y >> 16), sx, sy);
        if (!SC_DIAG_BIT(dbg, 1)) records[slot] = make_uint2(B.z, B.w | (mask << 28));
    }
    walk_big([&](int b, const Rect& br, int sx, int sy, unsigned bd, unsigned bi) {
        const unsigned slot = gbase[b] + atomicAdd(&hist[b], 1u);
        if (!SC_DIAG_BIT(dbg, 1)) records[slot] = make_uint2(bd, bi | (tile_mask(br.x0, br.x1, br.y0, br.y1, sx, sy) << 28));
    });
}

// ---- pass 3: per-super-tile sort + emit ---------------------------------------------------------------
constexpr int SS_THREADS = 512;
constexpr int SS_WAVES = SS_THREADS / 64;
constexpr int SS_MAX_CAP = 3584;                 // records one workgroup sorts in LDS (3 workgroups per CU); longer buckets are SPLIT first
constexpr int SS_RPT = SS_MAX_CAP / SS_THREADS;  // records per thread held in registers (7)
constexpr int SS_NC = SS_THREADS;                // coarse bins of the two-level interpolation (one per thread)
// Frames whose buckets are all small (S-100k: ~290 records per super-tile; a set of actors) are sorted by 128-thread
// workgroups: the 512-thread form ran 8 waves through every phase for half a record per thread (30 us at S-100k for
// 0.7 M records, 2400 workgroups in 2.3 rounds); two waves per bucket are all resident at once.
constexpr int SS_SMALL_THREADS = 128;
constexpr int SS_SMALL_RPT = 8;
constexpr int SS_SMALL_CAP = SS_SMALL_THREADS * SS_SMALL_RPT;      // 1024 records
constexpr int64_t BIG_MAX_SUPER = 220000;        // largest super-tile the split path takes (limits its LDS tables)

// A SEGMENT is what one workgroup sorts: a whole super-tile's bucket, or one depth range of an oversized one.
// tb[k] = entries of tile k (of the 2x2 super-tile) that precede this segment in the tile's list.
struct Segment { int start, n, sb, heavy; int tb[4]; };

__device__ __forceinline__ unsigned long long rec_key60(uint2 r) {      // (depth bits, flat id): the sort key
    return ((unsigned long long)r.x << 28) | (unsigned long long)(r.y & ID_MASK);
}

// Emits the per-tile lists of one segment from its records sorted on (depth, id) in LDS.
// Stable filter per tile: rank of record i in tile k's list = number of records j < i with mask bit k;
// computed per (round, wave) with ballots, bases by a tiny scan over the (round, wave) table.
// `order` (nullable): when given, the sorted sequence is S[order[0]], S[order[1]], ... (the
// interpolation sort keeps 2-byte ranks instead of a second copy of the keys: 14 instead of 20 B of LDS
// per record).  tb: see Segment.  totals (nullable, LDS [4]): receives the segment's per-tile counts.
template <int THREADS>
__device__ __forceinline__ void emit_tiles(const unsigned long long* __restrict__ S,
                                           const unsigned short* __restrict__ order, int n, int sb, const Geo& g,
                                           const int32_t* __restrict__ offsets, int n_tiles_total,
                                           int64_t n_isects, int tile_bits, const int* tb, unsigned long long* table,
                                           int64_t* __restrict__ isect_ids, int32_t* __restrict__ flatten_ids,
                                           unsigned* totals SC_DIAG_PARAM(dbg)) {
    constexpr int WAVES = THREADS / 64;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cam = sb / g.ST, srem = sb - cam * g.ST;
    const int sy = srem / g.stw, sx = srem - sy * g.stw;
    const int ntile = g.ss ? 4 : 1;
    const int rounds = (n + THREADS - 1) / THREADS;
    // pass A: per (round, wave) packed counts (16 bits per tile; a segment has <= SS_MAX_CAP records)
    for (int r = 0; r < rounds; ++r) {
        const int i = r * THREADS + t;
        const unsigned m = (i < n) ? (unsigned)((S[order ? order[i] : i] >> 28) & 0xfu) : 0u;
        unsigned long long packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            packed |= (unsigned long long)__popcll(__ballot((m >> k) & 1u)) << (16 * k);
        if (lane == 0) table[r * WAVES + wave] = packed;
    }
    __syncthreads();
    // exclusive scan of the table (<= rounds * WAVES entries) by the first wave
    if (wave == 0) {
        const int cnt = rounds * WAVES;
        unsigned long long carry = 0;
        for (int b0 = 0; b0 < cnt; b0 += 64) {
            const int i = b0 + lane;
            const unsigned long long v = (i < cnt) ? table[i] : 0ull;
            const unsigned long long incl = (unsigned long long)sc_wave_incl_scan64((long long)v);
            if (i < cnt) table[i] = carry + incl - v;
            carry += (unsigned long long)__shfl((long long)incl, 63, 64);
        }
        if (totals && lane < 4) totals[lane] = (unsigned)((carry >> (16 * lane)) & 0xffffu);
    }
    __syncthreads();
    // tile bases
    long long hi_key[4];
    int tbase[4], tend[4];
    bool tok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int tx = (sx << g.ss) + (k & 1), ty = (sy << g.ss) + (k >> 1);
        tok[k] = (k < ntile) && tx < g.tile_width && ty < g.tile_height;
        const int tile = ty * g.tile_width + tx;
        const int tflat = cam * g.T + tile;
        tbase[k] = tok[k] ? offsets[tflat] + tb[k] : 0;
        // end of this tile's list: a write is dropped rather than allowed past it, so that records
        // whose masks disagree with the counts (corrupt input, diagnostic skips) cannot go out of bounds
        tend[k] = tok[k] ? ((tflat + 1 < n_tiles_total) ? offsets[tflat + 1] : (int)n_isects) : 0;
        hi_key[k] = ((long long)cam << (32 + tile_bits)) | ((long long)tile << 32);
    }
    if (SC_DIAG_BIT(dbg, 1)) return;
    for (int r = 0; r < rounds; ++r) {
        const int i = r * THREADS + t;
        const unsigned long long key = (i < n) ? S[order ? order[i] : i] : 0ull;
        const unsigned m = (i < n) ? (unsigned)((key >> 28) & 0xfu) : 0u;
        const unsigned long long base = table[r * WAVES + wave];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long bal = __ballot((m >> k) & 1u);
            if (((m >> k) & 1u) && tok[k]) {
                const int pos = tbase[k] + (int)((base >> (16 * k)) & 0xffffu) + __popcll(bal & sc_lanemask_lt());
                if (pos < tend[k]) {
                    if (isect_ids && !SC_DIAG_BIT(dbg, 16)) isect_ids[pos] = hi_key[k] | (long long)(key >> 32);
                    flatten_ids[pos] = (int32_t)((unsigned)key & ID_MASK);
                }
            }
        }
    }
}

// ---- PULL route: a bucket's workgroup gathers its own records ---------------------------------------------------
// LDS words the gather needs: seg_start[nrows], seg_off[nrows + 1], the hit counter
__host__ __device__ inline size_t pull_lds_words(int nrows) { return (size_t)(2 * nrows + 4); }

__device__ __forceinline__ unsigned pull_tile_mask(int x0, int x1, int y0, int y1, int bx, int by, int ss) {
    // (x0, x1, y0, y1): the Gaussian's TILE rectangle; (bx, by): the super-tile.  0 = the rectangle misses the bucket
    if (!ss) return (x0 <= bx && bx < x1 && y0 <= by && by < y1) ? 1u : 0u;
    unsigned m = 0;
    const int tx = bx << 1, ty = by << 1;
    const bool cx0 = tx >= x0 && tx < x1, cx1 = tx + 1 >= x0 && tx + 1 < x1;
    if (ty >= y0 && ty < y1) { if (cx0) m |= 1u; if (cx1) m |= 2u; }
    if (ty + 1 >= y0 && ty + 1 < y1) { if (cx0) m |= 4u; if (cx1) m |= 8u; }
    return m;
}

// Every thread of the workgroup calls it.  store(slot, depth bits, id | mask << 28) is called once per record of
// super-tile bucket `sb`, slots 0 .. hits - 1 in arbitrary order; returns the number of hits.
//   pl   : LDS, pull_lds_words(pt.nrows) words
//   cand : LDS, cand_cap words (a multiple of 64): the payload indices of one chunk of candidates
// Window rows -> runs of the payload (two reads of the key starts per row), exclusive scan of the run lengths, then
// the runs are spelled out as candidate indices in LDS so that the candidate loop has 64 busy lanes and all of a
// thread's payload loads in flight together (a run is ~17 candidates long on average).
template <int THREADS, typename F>
__device__ __forceinline__ int pull_bucket(const uint4* __restrict__ sorted, const int32_t* __restrict__ cstart,
                                           int64_t n_visible, const PullTab& pt, const Geo& g, int sb, unsigned* pl,
                                           unsigned* cand, int cand_cap, F&& store) {
    constexpr int WAVES = THREADS / 64, U = 4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cam = sb / g.ST, srem = sb - cam * g.ST;
    const int by = srem / g.stw, bx = srem - by * g.stw;
    const int nrows = pt.nrows;
    unsigned* seg_start = pl;
    unsigned* seg_off = pl + nrows;               // [nrows + 1]
    unsigned* hits = pl + 2 * nrows + 1;
    const int smax = max(g.stw, g.sth);
    for (int r = t; r < nrows; r += THREADS) {
        int c = 0, r0 = 0;
#pragma unroll
        for (int k = 1; k < PULL_MAX_CLS; ++k)
            if (k < g.ncls && r >= pt.row0[k]) { c = k; r0 = pt.row0[k]; }
        const int dy = r - r0;
        const int sz = c == g.ncls - 1 ? smax : pull_size(c);
        const int y = by - dy;
        unsigned st = 0, len = 0;
        if (y >= 0 && dy < sz) {
            const int rowkey = ((cam * g.ncls + c) * g.sth + y) * g.stw;
            const int a = cstart[rowkey + max(bx - sz + 1, 0)], b = cstart[rowkey + bx + 1];
            if (a >= 0 && b > a && (int64_t)b <= n_visible) { st = (unsigned)a; len = (unsigned)(b - a); }
        }
        seg_start[r] = st;
        seg_off[r] = len;
    }
    if (t == 0) *hits = 0u;
    __syncthreads();
    if (wave == 0) {                               // exclusive scan of the run lengths
        unsigned carry = 0;
        for (int b0 = 0; b0 < nrows; b0 += 64) {
            const int i = b0 + lane;
            const unsigned v = i < nrows ? seg_off[i] : 0u;
            const unsigned incl = (unsigned)sc_wave_incl_scan((int)v);
            if (i < nrows) seg_off[i] = carry + incl - v;
            carry += (unsigned)__shfl((int)incl, 63, 64);
        }
        if (lane == 0) seg_off[nrows] = carry;
    }
    __syncthreads();
    const int total = (int)seg_off[nrows];
    for (int c0 = 0; c0 < total; c0 += cand_cap) {
        const int cn = min(cand_cap, total - c0);
        for (int r = wave; r < nrows; r += WAVES) {           // runs round-robin over the waves, 64 candidates per step
            const int off = (int)seg_off[r], end = (int)seg_off[r + 1];
            const int lo = max(off, c0), hi = min(end, c0 + cn);
            const unsigned st = seg_start[r];
            for (int i = lo + lane; i < hi; i += 64) cand[i - c0] = st + (unsigned)(i - off);
        }
        __syncthreads();
        for (int base = 0; base < cn; base += THREADS * U) {   // wave-uniform trip count (ballots inside)
            uint4 pay[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * THREADS + t;
                pay[u] = i < cn ? sorted[cand[i]] : make_uint4(0u, 0u, 0u, 0u);       // (0, 0, 0, 0): an empty rectangle
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned m = pull_tile_mask((int)(pay[u].x & 0xffffu), (int)(pay[u].x >> 16), (int)(pay[u].y & 0xffffu),
                                                  (int)(pay[u].y >> 16), bx, by, g.ss);
                const unsigned long long bal = __ballot(m != 0u);
                if (bal) {
                    unsigned b = 0;
                    if (lane == 0) b = atomicAdd(hits, (unsigned)__popcll(bal));
                    b = (unsigned)__shfl((int)b, 0, 64);
                    if (m) store(b + (unsigned)__popcll(bal & sc_lanemask_lt()), pay[u].z, (pay[u].w & ID_MASK) | (m << 28));
                }
            }
        }
        __syncthreads();
    }
    return (int)*hits;
}

// LDS of the sort: [B: cap u64][order: cap u16][boff: cap + 2 u32][coarse: NC u32][table][totals 4 u32]
// (14 B per record + 2.6 KiB: three workgroups per CU up to cap = 3584)
// (`threads` = the workgroup size of the instantiation: SS_THREADS, or SS_SMALL_THREADS for frames of small buckets)
__host__ __device__ inline size_t sort_lds_bytes(int cap, int threads = SS_THREADS) {
    const size_t table = (size_t)((cap + threads - 1) / threads + 1) * (threads / 64) * 8 + 64;
    return (size_t)cap * 10 + (size_t)((cap + 4) & ~1) * 4 + (size_t)threads * 4 + table + 16;
}

// Two-level interpolation sort of one segment (n <= cap <= SS_MAX_CAP records) on the 60-bit key
// (depth bits, flat id), then the stable per-tile emit.
//   level 1: NC coarse bins by a MONOTONE linear map of the key over [lo, hi] (double precision: the
//            map must be exactly non-decreasing in the key -- a correctly rounded u64 -> f64 conversion,
//            one multiply, one truncation);
//   level 2: a coarse bin that received c records is cut into c FINE buckets by the same map restricted to
//            the bin -- a piecewise-linear equalisation, so a depth distribution with narrow clusters and
//            far outliers (a facade plus sky, the normal case in a street scene) still averages about one
//            record per fine bucket (round 1's single-level map piled such buckets hundreds deep and sent
//            them to a slow radix fallback);
//   counting sort into the n fine buckets, then every record is ranked inside its bucket by counting
//   smaller keys.  Whatever the distribution the result is the exact order; only the cost of the ranking
//   loop depends on it (equal depths are separated by the id bits of the key).
struct PullCtx { const uint4* sorted; const int32_t* cstart; int64_t n_visible; const PullTab* pt; };

// PULL: the records are gathered from the spatially sorted payload (pull_bucket) instead of read from `recs`; `n` is then
// the bucket's expected size (from the count phase) and only bounds the result.
template <int THREADS, int RPT, bool PULL = false>
__device__ __forceinline__ void sort_segment(
    const uint2* __restrict__ recs, int n, int sb, const int* tb, int cap, const Geo& g,
    const int32_t* __restrict__ offsets, int n_tbuckets, int64_t n_isects, int tile_bits,
    unsigned char* smem, int64_t* __restrict__ isect_ids, int32_t* __restrict__ flatten_ids, unsigned* totals_out,
    const PullCtx& pc SC_DIAG_PARAM(dbg)) {
    constexpr int WAVES = THREADS / 64, NC = THREADS;          // coarse bins: one per thread
    unsigned long long* B = reinterpret_cast<unsigned long long*>(smem);
    unsigned short* order = reinterpret_cast<unsigned short*>(B + cap);      // cap is a multiple of 256
    unsigned* boff = reinterpret_cast<unsigned*>(order + cap);              // [n + 1] fine-bucket counters
    unsigned* coarse = boff + ((cap + 4) & ~1);       // [NC] count in the low, first fine bucket in the high half
    unsigned long long* table = reinterpret_cast<unsigned long long*>(coarse + NC);
    unsigned* totals = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(table) +
                                                   (size_t)((cap + THREADS - 1) / THREADS + 1) * WAVES * 8 + 64);
    __shared__ unsigned long long red_lo[WAVES], red_hi[WAVES];
    __shared__ unsigned red_sum[WAVES];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    // the segment's records are read ONCE (from global memory, all loads of a thread in flight together; or gathered
    // into B by the pull) and stay in registers through the min/max, both counting passes and the scatter
    uint2 rec[RPT];
    if (PULL) {
        // the candidate indices and the run table live where order[] / boff[] will be (6 B per record, free until the sort)
        const int pw = (int)pull_lds_words(pc.pt->nrows);
        const int cand_cap = ((cap * 6) / 4 - pw) & ~63;        // >= 64: the host keeps cap >= 1024 on the pull route
        unsigned* cand = reinterpret_cast<unsigned*>(order);
        const int nh = pull_bucket<THREADS>(pc.sorted, pc.cstart, pc.n_visible, *pc.pt, g, sb, cand + cand_cap, cand, cand_cap,
                                            [&](unsigned slot, unsigned depth, unsigned idw) {
                                                if (slot < (unsigned)cap) B[slot] = ((unsigned long long)depth << 32) | idw;
                                            });
        n = min(min(nh, n), cap);         // (nh == n unless the inputs changed between the count and the sort phase)
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int i = t + k * THREADS;
            const unsigned long long v = (i < n) ? B[i] : 0ull;
            rec[k] = make_uint2((unsigned)(v >> 32), (unsigned)v);
        }
        __syncthreads();                  // cand[] is dead, B is in registers
    } else {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int i = t + k * THREADS;
            rec[k] = (i < n) ? recs[i] : make_uint2(0u, 0u);
        }
    }
    for (int i = t; i <= n; i += THREADS) boff[i] = 0;
    coarse[t] = 0u;
    unsigned long long lo = ~0ull, hi = 0ull;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        if (t + k * THREADS < n) {
            const unsigned long long K = rec_key60(rec[k]);
            lo = K < lo ? K : lo;
            hi = K > hi ? K : hi;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long l2 = (unsigned long long)__shfl_xor((long long)lo, o, 64);
        const unsigned long long h2 = (unsigned long long)__shfl_xor((long long)hi, o, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if (lane == 0) { red_lo[wave] = lo; red_hi[wave] = hi; }
    __syncthreads();                      // also: boff / coarse zeroed
    lo = red_lo[0]; hi = red_hi[0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) {
        lo = red_lo[w] < lo ? red_lo[w] : lo;
        hi = red_hi[w] > hi ? red_hi[w] : hi;
    }
    if (SC_DIAG_BIT(dbg, 4)) return;      // diagnostic: price the load + min / max alone (nothing is emitted)
    const double s1 = (double)NC / ((double)(hi - lo) + 1.0);
    // key -> (coarse bin, position inside the bin in [0, 1]); monotone in the key
    auto level1 = [&](unsigned long long K, float& frac) -> int {
        const double p = (double)(K - lo) * s1;
        int c = (int)p;
        c = c < NC - 1 ? c : NC - 1;
        frac = (float)(p - (double)c);    // p - c is exact; (float) rounds monotonically
        return c;
    };
    // (the kernel needs <= 80 VGPRs for three workgroups per CU: the coarse bin is recomputed in the second
    //  counting pass, the fine bucket is kept from there for the scatter)
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        if (t + k * THREADS < n) {
            float frac;
            atomicAdd(&coarse[level1(rec_key60(rec[k]), frac)], 1u);
        }
    }
    __syncthreads();
    {   // exclusive scan of the coarse counts: thread t owns bin t
        const unsigned c = coarse[t];              // <= cap <= SS_MAX_CAP: fits 16 bits
        const unsigned incl = (unsigned)sc_wave_incl_scan((int)c);
        if (lane == 63) red_sum[wave] = incl;
        __syncthreads();
        unsigned run = incl - c;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) if (w < wave) run += red_sum[w];
        coarse[t] = c | (run << 16);
    }
    __syncthreads();
    if (SC_DIAG_BIT(dbg, 8)) return;      // diagnostic: ... and the coarse counting pass + scan
    auto fine_of = [&](unsigned long long K) -> int {
        float frac;
        const unsigned cs = coarse[level1(K, frac)];
        const int cnt = (int)(cs & 0xffffu);
        int f = (int)(frac * (float)cnt);
        f = f < cnt - 1 ? f : cnt - 1;
        return (int)(cs >> 16) + (f > 0 ? f : 0);
    };
    int fj[RPT];                            // the fine bucket of each of this thread's records (kept for the scatter)
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        fj[k] = 0;
        if (t + k * THREADS < n) {
            fj[k] = fine_of(rec_key60(rec[k]));
            atomicAdd(&boff[fj[k]], 1u);
        }
    }
    __syncthreads();
    {   // exclusive scan of the n fine counters: thread t owns `per` consecutive ones (per is ODD: the
        // lanes' strides then hit 32 distinct banks instead of two)
        const int per = ((n + THREADS - 1) / THREADS) | 1;
        const int beg = min(t * per, n), end = min(beg + per, n);
        unsigned sum = 0;
        for (int i = beg; i < end; ++i) sum += boff[i];
        const unsigned incl = (unsigned)sc_wave_incl_scan((int)sum);
        __syncthreads();                  // red_sum is reused
        if (lane == 63) red_sum[wave] = incl;
        __syncthreads();
        unsigned run = incl - sum;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) if (w < wave) run += red_sum[w];
        for (int i = beg; i < end; ++i) {      // counts -> bucket starts (used as running cursors)
            const unsigned c = boff[i];
            boff[i] = run;
            run += c;
        }
    }
    __syncthreads();
    if (SC_DIAG_BIT(dbg, 32)) return;     // diagnostic: ... and the fine counting pass + scan
    // scatter: after this pass boff[j] is the END of fine bucket j (== start of j + 1)
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        if (t + k * THREADS < n) {
            const uint2 r = rec[k];
            const int f = fj[k];
            const unsigned slot = atomicAdd(&boff[f], 1u);
            B[slot] = ((unsigned long long)r.x << 32) | r.y;
            order[slot] = (unsigned short)f;          // the bucket of this position, for the ranking pass
        }
    }
    __syncthreads();
    if (SC_DIAG_BIT(dbg, 2)) return;
    // rank inside the fine bucket by (depth bits, flat id) -> order[rank] = position in B.  Consecutive
    // threads take consecutive positions, i.e. neighbouring buckets: the LDS reads stay close together.
    // The bucket of position p was left in order[p] by the scatter (2 bytes instead of re-deriving it from the
    // key: ~40 VALU ops of double-precision mapping per record); all of a thread's bucket ids are read before
    // any rank is written, because order[] is also the output.  (One thread per BUCKET ranking its keys in
    // registers was measured too: 35 us instead of 22 for this phase -- divergence.)
    unsigned short bj[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int p = t + k * THREADS;
        bj[k] = p < n ? order[p] : (unsigned short)0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int p = t + k * THREADS;
        if (p >= n) break;
        const unsigned long long kk = B[p] & KEY_MASK;
        const int j = bj[k];
        const unsigned beg = j > 0 ? boff[j - 1] : 0u, end = boff[j];
        unsigned r = beg;
        for (unsigned q = beg; q < end; ++q) r += ((B[q] & KEY_MASK) < kk) ? 1u : 0u;
        order[r] = (unsigned short)p;
    }
    __syncthreads();
    emit_tiles<THREADS>(B, order, n, sb, g, offsets, n_tbuckets, n_isects, tile_bits, tb, table, isect_ids,
                           flatten_ids, totals SC_DIAG_ARG(dbg));
    if (totals_out) {
        __syncthreads();
        if (t < 4) totals_out[t] = totals[t];
    }
}

// Oversized segment whose keys could not be split by depth (a heavy histogram bin: thousands of records within
// 1/1024 of the super-tile's key range): exact, slow.  Rank every record among all n by counting smaller keys
// straight from global memory, scatter into `dst` (the super-tile's original bucket, free by now), then emit
// chunk by chunk through LDS with running per-tile bases.
__device__ __forceinline__ void sort_heavy_segment(
    const uint2* __restrict__ src, uint2* __restrict__ dst, int n, int sb, const int* tb_in, int cap, const Geo& g,
    const int32_t* __restrict__ offsets, int n_tbuckets, int64_t n_isects, int tile_bits, unsigned char* smem,
    int64_t* __restrict__ isect_ids, int32_t* __restrict__ flatten_ids) {
    unsigned long long* B = reinterpret_cast<unsigned long long*>(smem);
    unsigned long long* table = reinterpret_cast<unsigned long long*>(smem + (size_t)cap * 8);
    __shared__ unsigned tot[4];
    __shared__ int tbr[4];
    const int t = threadIdx.x;
    for (int i = t; i < n; i += SS_THREADS) {
        const uint2 r = src[i];
        const unsigned long long K = rec_key60(r);
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += rec_key60(src[j]) < K ? 1 : 0;
        dst[rank] = r;
    }
    if (t < 4) tbr[t] = tb_in[t];
    __threadfence();
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += cap) {
        const int m = min(cap, n - c0);
        for (int i = t; i < m; i += SS_THREADS) {
            // (agent-scope load: bypasses this CU's L1, which may hold a line of `dst` from before the scatter)
            const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(dst + c0 + i),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            B[i] = (v << 32) | (v >> 32);           // uint2 {x = depth, y = mask | id} is x in the low word
        }
        __syncthreads();
        int tbl[4] = {tbr[0], tbr[1], tbr[2], tbr[3]};
        emit_tiles<SS_THREADS>(B, nullptr, m, sb, g, offsets, n_tbuckets, n_isects, tile_bits, tbl, table, isect_ids,
                               flatten_ids, tot SC_DIAG_ARG(0));
        __syncthreads();
        if (t < 4) tbr[t] += (int)tot[t];
        __syncthreads();
    }
}

// One launch sorts everything: blocks [0, seg_bound) take the segments big_split_kernel produced (none in the
// common case: those blocks return at once), blocks [seg_bound, seg_bound + n_sbuckets) the super-tiles whose
// bucket fits the LDS capacity.
template <int THREADS, int RPT, bool PULL>
__global__ __launch_bounds__(THREADS, 6) void super_sort_kernel(
    const uint2* __restrict__ records, uint2* __restrict__ records_rw, const uint2* __restrict__ temp,
    const Segment* __restrict__ segs, const unsigned* __restrict__ n_segs, int seg_bound,
    const int32_t* __restrict__ soffsets, int n_sbuckets, int n_tbuckets,
    Geo g, const int32_t* __restrict__ offsets, const int64_t* __restrict__ meta, int64_t capacity,
    int64_t rec_capacity, int64_t super_capacity, int tile_bits, int cap,
    int64_t* __restrict__ isect_ids, int32_t* __restrict__ flatten_ids,
    const uint4* __restrict__ sorted, const int32_t* __restrict__ cstart, const int64_t* __restrict__ n_visible,
    PullTab pt SC_DIAG_PARAM(dbg)) {
    extern __shared__ __align__(16) unsigned char smem[];
    // the SAME three comparisons in every kernel of the sort phase and in the host wrapper
    // (rendering._bin_launch_ran): a launch either runs in full or not at all
    if (meta[0] > capacity || meta[2] > rec_capacity || meta[3] > super_capacity) return;
    __shared__ int tb_s[4];
    PullCtx pc;
    pc.sorted = sorted; pc.cstart = cstart; pc.n_visible = PULL ? n_visible[0] : 0; pc.pt = &pt;
    if (THREADS == SS_THREADS && (int)blockIdx.x < seg_bound) {      // (the small form is launched without segments)
        if (blockIdx.x >= *n_segs) return;
        const Segment* sg = segs + blockIdx.x;
        if (threadIdx.x < 4) tb_s[threadIdx.x] = sg->tb[threadIdx.x];
        const int start = sg->start, sn = sg->n, ssb = sg->sb;
        __syncthreads();
        if (sg->heavy)
            sort_heavy_segment(temp + start, records_rw + start, sn, ssb, tb_s, cap, g, offsets, n_tbuckets,
                               meta[0], tile_bits, smem, isect_ids, flatten_ids);
        else
            sort_segment<SS_THREADS, SS_RPT, false>(temp + start, sn, ssb, tb_s, cap, g, offsets, n_tbuckets, meta[0], tile_bits,
                                                    smem, isect_ids, flatten_ids, nullptr, pc SC_DIAG_ARG(dbg));
        return;
    }
    int sb = (int)blockIdx.x - seg_bound;
    if (PULL) {
        // neighbouring buckets gather overlapping windows of the payload: blocks go round-robin over the 8 XCDs, so XCD x
        // takes the x-th eighth of the buckets (a band of super-tile rows) and its L2 holds that band's part of the payload
        const int per = (n_sbuckets + 7) >> 3;
        sb = (sb & 7) * per + (sb >> 3);
        if (sb >= n_sbuckets) return;
    }
    const int s = soffsets[sb];
    const int e = (sb + 1 < n_sbuckets) ? soffsets[sb + 1] : (int)meta[2];
    const int n = e - s;
    if (n <= 0 || n > cap) return;          // n > cap: big_split_kernel has cut this bucket into segments
    if (threadIdx.x < 4) tb_s[threadIdx.x] = 0;
    __syncthreads();
    sort_segment<THREADS, RPT, PULL>(records + s, n, sb, tb_s, cap, g, offsets, n_tbuckets, meta[0], tile_bits, smem, isect_ids,
                                     flatten_ids, nullptr, pc SC_DIAG_ARG(dbg));
}

// PULL route, oversized buckets (n > cap): their records are gathered into the records buffer -- the slice the scatter
// route would have filled -- and big_split_kernel then cuts them into segments as before.  One workgroup per bucket.
constexpr int BP_THREADS = 1024;
constexpr int BP_CAND = 8192;                     // candidate indices per chunk (32 KB of LDS)
__global__ __launch_bounds__(BP_THREADS) void big_pull_kernel(
    const uint4* __restrict__ sorted, const int32_t* __restrict__ cstart, const int64_t* __restrict__ n_visible, PullTab pt,
    Geo g, const int32_t* __restrict__ soffsets, int n_sbuckets, const int64_t* __restrict__ meta, int64_t capacity,
    int64_t rec_capacity, int64_t super_capacity, int cap, uint2* __restrict__ records) {
    extern __shared__ unsigned lds[];
    if (meta[0] > capacity || meta[2] > rec_capacity || meta[3] > super_capacity) return;
    const int sb = (int)blockIdx.x;
    const int s = soffsets[sb];
    const int e = (sb + 1 < n_sbuckets) ? soffsets[sb + 1] : (int)meta[2];
    const int n = e - s;
    if (n <= cap || (int64_t)e > rec_capacity) return;
    uint2* dst = records + s;
    pull_bucket<BP_THREADS>(sorted, cstart, n_visible[0], pt, g, sb, lds + BP_CAND, lds, BP_CAND,
                            [&](unsigned slot, unsigned depth, unsigned idw) {
                                if (slot < (unsigned)n) dst[slot] = make_uint2(depth, idw);
                            });
}

// ---- isect_ids on demand ------------------------------------------------------------------------------------
// isect_ids[k] = (camera << (32 + tile_bits)) | (tile << 32) | depth bits of the Gaussian at position k of the
// sorted lists: everything needed is in flatten_ids, isect_offsets and depths, so the 8 B x I key array need not
// be written by the sort at all (nothing on the reference's path reads it: street_crafter_amd/lazy.py).  One
// workgroup per (camera, tile).
__global__ __launch_bounds__(256) void isect_ids_rebuild_kernel(
    const int32_t* __restrict__ flatten_ids, const int32_t* __restrict__ offsets, const float* __restrict__ depths,
    int n_tiles_total, int tiles_per_cam, int tile_bits, int64_t CN, int64_t n_isects, int64_t* __restrict__ isect_ids) {
    const int tflat = blockIdx.x;
    const int cam = tflat / tiles_per_cam, tile = tflat - cam * tiles_per_cam;
    int s = offsets[tflat];
    int e = (tflat + 1 < n_tiles_total) ? offsets[tflat + 1] : (int)n_isects;
    s = min(max(s, 0), (int)n_isects);
    e = min(max(e, s), (int)n_isects);
    const long long hi = ((long long)cam << (32 + tile_bits)) | ((long long)tile << 32);
    for (int k = s + (int)threadIdx.x; k < e; k += 256) {
        const int g = flatten_ids[k];
        const unsigned d = ((unsigned)g < (unsigned)CN) ? __float_as_uint(depths[g]) : 0u;
        isect_ids[k] = hi | (long long)d;
    }
}

// ---- oversized buckets: cut into depth ranges that fit the LDS sort ---------------------------------------
// One workgroup per oversized super-tile (n > cap records; grid-stride over the super-tiles):
//   1. min / max of the 60-bit key; histogram of the records over BS_NB bins of a monotone linear map;
//   2. consecutive bins are grouped into RANGES of at most cap records (greedy on the prefix sums; a bin
//      that alone exceeds cap / 2 becomes a range of its own, `heavy` if it exceeds cap);
//   3. the records are copied into `temp` grouped by range (order inside a range is arbitrary: the segment
//      sort fixes it), with, per range, the number of its records in each of the 4 tiles -- their prefix
//      sums are the tile bases the range's workgroup adds when it emits;
//   4. one Segment per range.
constexpr int BS_THREADS = 1024;
constexpr int BS_NB = 1024;                       // histogram bins == threads
constexpr int BS_MAX_RANGES = 256;                // ranges one oversized bucket may be cut into (LDS tables)
constexpr int BS_UNROLL = 8;                      // records a thread has in flight per step (the passes are latency-bound)

// f(record, valid) over the n records at `src`, BS_UNROLL independent loads per thread in flight
template <typename F>
__device__ __forceinline__ void for_records(const uint2* __restrict__ src, int n, F&& f) {
    for (int i0 = threadIdx.x; i0 < n; i0 += BS_THREADS * BS_UNROLL) {
        uint2 v[BS_UNROLL];
#pragma unroll
        for (int u = 0; u < BS_UNROLL; ++u) {
            const int i = i0 + u * BS_THREADS;
            v[u] = i < n ? src[i] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < BS_UNROLL; ++u)
            if (i0 + u * BS_THREADS < n) f(v[u]);
    }
}

__global__ __launch_bounds__(BS_THREADS, 8) void big_split_kernel(
    const uint2* __restrict__ records, uint2* __restrict__ temp, Segment* __restrict__ segs,
    unsigned* __restrict__ n_segs, int seg_bound, const int32_t* __restrict__ soffsets, int n_sbuckets,
    const int64_t* __restrict__ meta, int64_t capacity, int64_t rec_capacity, int64_t super_capacity, int cap
    SC_DIAG_PARAM(dbg)) {
    if (meta[0] > capacity || meta[2] > rec_capacity || meta[3] > super_capacity) return;
    // The keys of a street-scene bucket sit in a few narrow depth bands, so many lanes of a wave hit the SAME
    // few counters and an LDS atomic instruction serialises identical addresses: every bin counter has 8
    // replicas picked by lane & 7, the per-range tile counters 4.  (Measured on the street scene, 1 M Gaussians,
    // ~6 M of 8.8 M records in oversized buckets: the kernel takes ~100 us whatever the counters look like --
    // it is three latency-bound passes over the buckets, one workgroup each; wave-aggregated adds with ballots
    // instead of per-lane atomics made the copy pass 6x slower.)
    constexpr int REP = 8;
    __shared__ unsigned hist[BS_NB * REP];         // replicated bin counts; hist[b * REP] later: range of bin b
    constexpr int TREP = 4;
    __shared__ unsigned rcnt[BS_MAX_RANGES], rtile[BS_MAX_RANGES][4][TREP], rstart[BS_MAX_RANGES], rcur[BS_MAX_RANGES];
    __shared__ unsigned long long red_lo[BS_THREADS / 64], red_hi[BS_THREADS / 64];
    __shared__ unsigned wtot[BS_THREADS / 64], wheavy[BS_THREADS / 64];
    __shared__ unsigned n_ranges_s, seg_base_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // Consecutive LIGHT bins (<= cap / 4 records) are grouped while their running total stays inside one multiple of `target`
    // = 3/4 cap: a group holds < target + cap / 4 = cap records.  (Round 4: until then target = cap / 2 and bins up to that size
    // were light -- segments half full on average; fuller segments are fewer sort workgroups: super_sort 119 -> 107 us on the
    // street scene at 1 M, 327 -> 287 us at 3 M, profiles/r04_big_split_sampled_ab.txt.)
    const int target = (cap * 3) / 4, light_max = cap / 4;
    for (int sb = blockIdx.x; sb < n_sbuckets; sb += gridDim.x) {
        const int s = soffsets[sb];
        const int e = (sb + 1 < n_sbuckets) ? soffsets[sb + 1] : (int)meta[2];
        const int n = e - s;
        if (n <= cap) continue;                    // workgroup-uniform
        const uint2* src = records + s;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < REP; ++k) hist[t * REP + k] = 0;
        unsigned long long lo = ~0ull, hi = 0ull;
        for_records(src, n, [&](uint2 rc) {
            const unsigned long long K = rec_key60(rc);
            lo = K < lo ? K : lo;
            hi = K > hi ? K : hi;
        });
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned long long l2 = (unsigned long long)__shfl_xor((long long)lo, o, 64);
            const unsigned long long h2 = (unsigned long long)__shfl_xor((long long)hi, o, 64);
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        if (lane == 0) { red_lo[wave] = lo; red_hi[wave] = hi; }
        __syncthreads();
        lo = red_lo[0]; hi = red_hi[0];
        for (int w = 1; w < BS_THREADS / 64; ++w) {
            lo = red_lo[w] < lo ? red_lo[w] : lo;
            hi = red_hi[w] > hi ? red_hi[w] : hi;
        }
        if (SC_DIAG_BIT(dbg, 8)) continue;          // diagnostic: price the min / max pass alone (no segments are made)
        const double sc = (double)BS_NB / ((double)(hi - lo) + 1.0);
        auto bin_of = [&](unsigned long long K) -> int {
            const int b = (int)((double)(K - lo) * sc);
            return b < BS_NB - 1 ? b : BS_NB - 1;
        };
        for_records(src, n, [&](uint2 rc) { atomicAdd(&hist[bin_of(rec_key60(rc)) * REP + (lane & (REP - 1))], 1u); });
        __syncthreads();
        if (SC_DIAG_BIT(dbg, 16)) continue;         // diagnostic: ... and the histogram pass
        // thread t owns bin t.  light = exclusive prefix of the light bins' counts, H = heavy bins before t.
        unsigned c = 0;
#pragma unroll
        for (int k = 0; k < REP; ++k) c += hist[t * REP + k];
        const bool heavy = c > (unsigned)light_max;
        const unsigned lc = heavy ? 0u : c;
        const unsigned incl = (unsigned)sc_wave_incl_scan((int)lc);
        const unsigned hincl = (unsigned)sc_wave_incl_scan(heavy ? 1 : 0);
        if (lane == 63) { wtot[wave] = incl; wheavy[wave] = hincl; }
        __syncthreads();
        unsigned light = incl - lc, H = hincl - (heavy ? 1u : 0u);
        for (int w = 0; w < BS_THREADS / 64; ++w)
            if (w < wave) { light += wtot[w]; H += wheavy[w]; }
        // sparse, monotone range id: light bins floor(light / target) + 2 H, a heavy bin the odd id after them
        const unsigned rid = light / (unsigned)target + 2u * H + (heavy ? 1u : 0u);
        // dense id = number of distinct rids among the non-empty bins up to this one, minus one.
        // The rid of the nearest non-empty bin to the LEFT comes from a running-maximum scan (rids are
        // monotone in the bin index; a serial walk over the mostly empty histogram cost tens of microseconds).
        const int mine = c ? (int)rid : -1;
        int run_max = mine;                        // inclusive running maximum inside the wave
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const int o = __shfl_up(run_max, dlt, 64);
            if (lane >= dlt) run_max = max(run_max, o);
        }
        int left = __shfl_up(run_max, 1, 64);      // exclusive: bins to the left inside the wave
        if (lane == 0) left = -1;
        __syncthreads();
        if (lane == 63) wheavy[wave] = (unsigned)(run_max + 1);     // (+1: stored unsigned, -1 -> 0)
        __syncthreads();
        for (int w = 0; w < BS_THREADS / 64; ++w)
            if (w < wave) left = max(left, (int)wheavy[w] - 1);
        const bool first = c && left != (int)rid;
        const unsigned fincl = (unsigned)sc_wave_incl_scan(first ? 1 : 0);
        __syncthreads();
        if (lane == 63) wtot[wave] = fincl;
        for (int r = t; r < BS_MAX_RANGES; r += BS_THREADS) {
            rcnt[r] = 0; rcur[r] = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int q = 0; q < TREP; ++q) rtile[r][k][q] = 0;
        }
        __syncthreads();
        unsigned dense = fincl;
        unsigned total_ranges = 0;
        for (int w = 0; w < BS_THREADS / 64; ++w) {
            if (w < wave) dense += wtot[w];
            total_ranges += wtot[w];
        }
        dense -= 1u;                                // this bin's range (meaningless for empty bins)
        hist[t * REP] = c ? dense : 0u;             // bin -> range
        if (t == 0) n_ranges_s = total_ranges;
        if (c) atomicAdd(&rcnt[dense], c);          // range totals from the bin totals
        __syncthreads();
        const int R = (int)n_ranges_s;             // <= 2 n / target + 1 <= BS_MAX_RANGES (the host bounds n)
        if (t == 0) {                               // R is small: serial prefix sum
            unsigned run = 0;
            for (int r = 0; r < R; ++r) { rstart[r] = run; run += rcnt[r]; }
            seg_base_s = atomicAdd(n_segs, (unsigned)R);
        }
        __syncthreads();
        // copy pass: records grouped by range; per range, how many of its records fall in each of the 4 tiles
        // (replicated counters again: a range's records are many lanes of every wave)
        uint2* dst = temp + s;
        if (!SC_DIAG_BIT(dbg, 4))        // diagnostic: skip the copy pass (segments then hold stale records; every consumer bounds-checks)
        for_records(src, n, [&](uint2 rc) {
            const unsigned r = hist[bin_of(rec_key60(rc)) * REP];
            const unsigned slot = rstart[r] + atomicAdd(&rcur[r], 1u);
            const unsigned m = rc.y >> 28;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((m >> k) & 1u) atomicAdd(&rtile[r][k][lane & (TREP - 1)], 1u);
            if (slot < (unsigned)n) dst[slot] = rc;      // (slot < n by construction; never trust it)
        });
        __syncthreads();
        if (t == 0) {                               // per-tile counts -> exclusive prefix over the ranges
            unsigned tr[4] = {0, 0, 0, 0};
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    unsigned v = 0;
#pragma unroll
                    for (int q = 0; q < TREP; ++q) v += rtile[r][k][q];
                    rtile[r][k][0] = tr[k];
                    tr[k] += v;
                }
            }
        }
        __syncthreads();
        const unsigned base = seg_base_s;
        for (int r = t; r < R; r += BS_THREADS) {
            if (base + r < (unsigned)seg_bound) {
                Segment sg;
                sg.start = s + (int)rstart[r]; sg.n = (int)rcnt[r]; sg.sb = sb; sg.heavy = (int)rcnt[r] > cap ? 1 : 0;
                sg.tb[0] = (int)rtile[r][0][0]; sg.tb[1] = (int)rtile[r][1][0];
                sg.tb[2] = (int)rtile[r][2][0]; sg.tb[3] = (int)rtile[r][3][0];
                segs[base + r] = sg;
            }
        }
    }
}

}  // namespace

// ---- host side --------------------------------------------------------------------------------
// count-phase workspace (handed to BOTH calls):
//   dgrid_t | dgrid_s | chist | ccursor | rcursor | nseg | scans_done | soffsets | cstart | smeta[2] | cmeta[2] | sorted uint4[CN]
//   (everything before soffsets is zeroed by the ONE memset of a frame; rcursor / nseg are the
//    scatter's bucket cursors and the segment counter of the sort phase, kept here so that the sort
//    phase needs no memset of its own: a second memset cost 5 us + a 6 us bubble)
// sort-phase workspace:
//   records uint2[rec_capacity] | (only when oversized buckets are provisioned for) temp uint2[rec_capacity] |
//   segments[seg_bound]
struct BinLayout {
    Geo g;
    int C, nt_cells, ns_cells, nsb, ntb;
    int nkeys;                 // pull route: (camera, size class, anchor) keys; 0 on the scatter route
    PullTab pt;
    size_t dgrid_t, dgrid_s, chist, ccursor, rcursor, nseg, scans_done, wstat, soffsets, cstart, smeta, cmeta, whint, sorted, total;
};

int g_sc_isect_pull = 0;       // sc_set_option "isect_pull": 1 = the pull route where the frame allows it, 0 (default) = the scatter route

// The pull route's size classes for a grid of stw x sth super-tiles (0 = the frame takes the scatter route: the key
// starts of a centre workgroup (4 B per key) and the window-row table must fit the LDS).
static int pull_classes(int C, int stw, int sth, PullTab* pt) {
    if (!g_sc_isect_pull) return 0;
    const int smax = stw > sth ? stw : sth;
    int n = 1;
    while (n < PULL_MAX_CLS && pull_size(n - 1) < smax) ++n;
    if ((size_t)C * n * stw * sth * 4 > 140 * 1024) return 0;
    int rows = 0;
    for (int c = 0; c < PULL_MAX_CLS + 1; ++c) pt->row0[c] = 0;
    for (int c = 0; c < n; ++c) {
        pt->row0[c] = rows;
        const int sz = c == n - 1 ? smax : pull_size(c);
        rows += sz < sth ? sz : sth;
    }
    for (int c = n; c <= PULL_MAX_CLS; ++c) pt->row0[c] = rows;
    pt->nrows = rows;
    if (rows > 1024) return 0;
    return n;
}

// (`pull`: -1 = as sc_set_option "isect_pull" and the frame's size decide; the layout is the same either way -- the
//  histogram / cursor arrays are sized for the larger of the two routes -- so that both calls of a frame agree on it)
static BinLayout bin_layout(int64_t CN, int C, int N, int tile_width, int tile_height) {
    BinLayout L;
    L.C = C;
    L.g.tile_width = tile_width; L.g.tile_height = tile_height; L.g.T = tile_width * tile_height;
    L.g.ss = 1;
    L.g.stw = (tile_width + 1) >> 1; L.g.sth = (tile_height + 1) >> 1; L.g.ST = L.g.stw * L.g.sth;
    L.g.N = N;
    L.ntb = C * L.g.T;
    L.nsb = C * L.g.ST;
    L.nt_cells = C * (tile_height + 1) * (tile_width + 1);
    L.ns_cells = C * (L.g.sth + 1) * (L.g.stw + 1);
    // the workspace is laid out for the pull route's key count whether or not the option is on (sizes must not depend on a
    // switch that can change between sc_isect_bin_workspace_bytes and the launch)
    PullTab tmp;
    const int keep = g_sc_isect_pull;
    g_sc_isect_pull = 1;
    const int ncls_max = pull_classes(C, L.g.stw, L.g.sth, &tmp);
    g_sc_isect_pull = keep;
    L.g.ncls = pull_classes(C, L.g.stw, L.g.sth, &L.pt);
    L.nkeys = L.g.ncls * L.nsb;
    const size_t nk = (size_t)(ncls_max > 0 ? ncls_max : 1) * L.nsb + 2;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += sc_align_up(bytes, 256); return at; };
    L.dgrid_t = take((size_t)L.nt_cells * 4);
    L.dgrid_s = take((size_t)L.ns_cells * 4);
    L.chist = take(nk * 4);
    L.ccursor = take(nk * 4);
    L.rcursor = take((size_t)L.nsb * 4);
    L.nseg = take(4);
    L.scans_done = take(4);
    L.wstat = take(16);                                   // maximum and sum of the rasterizer's work hints (zeroed per frame)
    L.soffsets = take((size_t)L.nsb * 4);
    L.cstart = take(nk * 4);                              // pull route: the key starts (nkeys + 1)
    L.smeta = take(16);
    L.cmeta = take(16);
    L.whint = take((size_t)L.ntb * 4);                    // the per-tile hints the order job files the tiles under
    L.sorted = take((size_t)(CN > 0 ? CN : 0) * 16);
    L.total = o;
    return L;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: done once per device, under a mutex
// (one process may drive several GPUs, and several host threads may enter the library)
#include <mutex>
static hipError_t bin_attrs_once() {
    static std::mutex mu;
    static bool done[64] = {false};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (done[dev]) return hipSuccess;
    const int a = hipFuncAttributeMaxDynamicSharedMemorySize;
    if ((e = hipFuncSetAttribute((const void*)bin_count_kernel<true, 8>, (hipFuncAttribute)a, 152 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)bin_count_kernel<true, 4>, (hipFuncAttribute)a, 152 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)bin_count_kernel<true, 2>, (hipFuncAttribute)a, 152 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)bin_count_kernel<true, 1>, (hipFuncAttribute)a, 152 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)bin_scatter_flat_kernel<FLAT_THREADS, 64>, (hipFuncAttribute)a, 152 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)bin_scatter_flat_kernel<FLAT_THREADS_SMALL, 16>, (hipFuncAttribute)a, 152 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)center_scatter_kernel, (hipFuncAttribute)a, 152 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)super_sort_kernel<SS_THREADS, SS_RPT, false>, (hipFuncAttribute)a, 100 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)super_sort_kernel<SS_THREADS, SS_RPT, true>, (hipFuncAttribute)a, 100 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)big_pull_kernel, (hipFuncAttribute)a, 64 * 1024)) != hipSuccess) return e;
    done[dev] = true;
    return hipSuccess;
}

// upper bound of the segments big_split_kernel can produce: 2 n / (cap / 2) + 1 per oversized bucket
static inline size_t seg_bound_for(int64_t n_records, int nsb) {
    return (size_t)(2 * (n_records > 0 ? n_records : 0) / (SS_MAX_CAP / 2)) + (size_t)nsb + 8;
}

static inline size_t count_lds_bytes(const BinLayout& L) {
    // (pull route: the key histogram as 16-bit halves)
    return (size_t)(L.nt_cells + L.ns_cells + (L.g.ncls ? (L.nkeys + 1) / 2 : L.nsb)) * 4;
}

extern "C" size_t sc_isect_bin_workspace_bytes(int64_t CN, int C, int tile_width, int tile_height,
                                               int64_t n_isects) {
    const int64_t nb = (int64_t)C * tile_width * tile_height;
    if (nb <= 0 || nb > BIN_MAX_TILES) return 256;
    const BinLayout L = bin_layout(CN, C, 1, tile_width, tile_height);
    if (n_isects < 0) return L.total;                                  // count-phase workspace
    // records + (for the split path) a second copy + the segment list; the wrapper cannot know here whether
    // oversized buckets will occur, so the size covers them: the extra bytes are only touched when they do
    return 2 * sc_align_up((size_t)n_isects * 8, 256) + sc_align_up(seg_bound_for(n_isects, L.nsb) * sizeof(Segment), 256) + 256;
}

extern "C" int sc_view_slots(void) { return SC_VIEW_SLOTS; }
extern "C" int sc_view_registry_words(void) { return SC_VIEW_REGISTRY_WORDS; }
extern "C" int sc_isect_bin_bucket_capacity(void) { return SS_MAX_CAP; }

extern "C" int sc_isect_bin_count(const float* means2d, const int32_t* radii, const float* depths, int C, int N,
                                  int tile_size,
                                  int tile_width, int tile_height, int32_t* tiles_per_gauss,
                                  int32_t* isect_offsets, int64_t* meta_dev, int64_t* meta_mirror,
                                  int64_t seq, void* count_workspace, size_t ws_bytes, const int32_t* tile_work,
                                  const float* viewmats, int32_t* view_registry, int32_t* tile_order,
                                  sc_stream_t stream) {
    if (C < 0 || N < 0 || tile_size <= 0 || tile_width <= 0 || tile_height <= 0) return SC_EINVAL;
    if (!meta_dev) return SC_EINVAL;
    const int64_t CN = (int64_t)C * N;
    const int64_t nb64 = (int64_t)C * tile_width * tile_height;
    if (nb64 > BIN_MAX_TILES || CN >= (1LL << 28)) return SC_EUNSUPPORTED;   // ids share a word with the mask
    hipStream_t s = sc_s(stream);
    if (CN == 0 || nb64 == 0) {
        if (nb64 > 0 && isect_offsets) SC_HIP(hipMemsetAsync(isect_offsets, 0, (size_t)nb64 * 4, s));
        SC_HIP(hipMemsetAsync(meta_dev, 0, 4 * sizeof(int64_t), s));
        if (meta_mirror) {
            hipLaunchKernelGGL(publish_meta_kernel, dim3(1), dim3(64), 0, s, (const int64_t*)meta_dev, meta_mirror, seq);
            SC_LAUNCH_CHECK();
        }
        if (tile_order && nb64 > 0) {         // nothing to rasterize, but the list must still name every tile
            const int n_total = sc_tile_order_len((int)nb64);
            hipLaunchKernelGGL(plain_order_kernel, dim3((unsigned)((n_total + 255) / 256)), dim3(256), 0, s, tile_order,
                               (int)nb64, n_total);
            SC_LAUNCH_CHECK();
        }
        return SC_OK;
    }
    if (!means2d || !radii || !depths || !tiles_per_gauss || !isect_offsets || !count_workspace) return SC_EINVAL;
    const BinLayout L = bin_layout(CN, C, N, tile_width, tile_height);
    if (ws_bytes < L.total) return SC_EWORKSPACE;
    // the grid-scan job keeps the tile grid in LDS, the centre pass 12 B per super-tile
    if ((size_t)L.nt_cells * 4 > 150 * 1024 || (size_t)L.nsb * 12 > 150 * 1024) return SC_EUNSUPPORTED;
    const bool local_grids = count_lds_bytes(L) <= 150 * 1024;
    unsigned char* ws = (unsigned char*)count_workspace;
    int* dgrid_t = (int*)(ws + L.dgrid_t);
    int* dgrid_s = (int*)(ws + L.dgrid_s);
    unsigned* chist = (unsigned*)(ws + L.chist);
    unsigned* ccursor = (unsigned*)(ws + L.ccursor);
    int32_t* soffsets = (int32_t*)(ws + L.soffsets);
    int64_t* cmeta = (int64_t*)(ws + L.cmeta);
    uint4* sorted = (uint4*)(ws + L.sorted);
    SC_HIP(hipMemsetAsync(ws, 0, L.soffsets, s));        // difference grids, chist, ccursor, rcursor, rflags
    SC_HIP(bin_attrs_once());
    const unsigned grid = (unsigned)((CN + BIN_GPB - 1) / BIN_GPB);
    // the count workgroups, then (with a work hint) one thread per tile that prepares the dispatch list's hints
    // Gaussians per thread of the count pass: 8 from ~800 k up (the flush of the per-workgroup grids dominates), fewer
    // for small inputs so that the pass still has ~100+ workgroups
    const int gpt = CN >= 786432 ? 8 : (CN >= 393216 ? 4 : (CN >= 196608 ? 2 : 1));
    const int n_count_blocks = (int)((CN + (int64_t)BIN_THREADS * gpt - 1) / ((int64_t)BIN_THREADS * gpt));
    const bool hints = tile_order && tile_work;
    const int n_hint_blocks = hints ? (L.ntb + BIN_THREADS - 1) / BIN_THREADS : 0;
    unsigned* whint = hints ? (unsigned*)(ws + L.whint) : nullptr;
    unsigned* wstat = (unsigned*)(ws + L.wstat);
    int32_t* slot_word = tile_order ? tile_order + (sc_tile_order_len(L.ntb) - 1) : nullptr;
    if (tile_order && !hints) SC_HIP(hipMemsetAsync(slot_word, 0, 4, s));      // (a hint block writes it otherwise)
#define SC_LAUNCH_COUNT(LOCALP, GPTP, LDSB)                                                                              \
    hipLaunchKernelGGL((bin_count_kernel<LOCALP, GPTP>), dim3((unsigned)(n_count_blocks + n_hint_blocks)),              \
                       dim3(BIN_THREADS), LDSB, s, means2d, radii, CN, L.g, (float)tile_size, C, tiles_per_gauss, dgrid_t, \
                       dgrid_s, chist, n_count_blocks, tile_work, whint, wstat, g_sc_raster_hint_blend, viewmats,           \
                       (unsigned*)view_registry, slot_word)
    if (local_grids) {
        const size_t ldsb = count_lds_bytes(L);
        if (gpt == 8) SC_LAUNCH_COUNT(true, 8, ldsb);
        else if (gpt == 4) SC_LAUNCH_COUNT(true, 4, ldsb);
        else if (gpt == 2) SC_LAUNCH_COUNT(true, 2, ldsb);
        else SC_LAUNCH_COUNT(true, 1, ldsb);
    } else {
        if (gpt == 8) SC_LAUNCH_COUNT(false, 8, 0);
        else if (gpt == 4) SC_LAUNCH_COUNT(false, 4, 0);
        else if (gpt == 2) SC_LAUNCH_COUNT(false, 2, 0);
        else SC_LAUNCH_COUNT(false, 1, 0);
    }
#undef SC_LAUNCH_COUNT
    SC_LAUNCH_CHECK();
    // tile grid -> isect_offsets + meta[0..1]; super-tile grid -> record offsets + meta[2..3]: two extra
    // blocks of the centre-scatter launch (see the kernel)
    ScanJobs jobs;
    jobs.j[0] = ScanJob{(const int*)dgrid_t, C, tile_width, tile_height, isect_offsets, meta_dev, 1};
    jobs.j[1] = ScanJob{(const int*)dgrid_s, C, L.g.stw, L.g.sth, soffsets, meta_dev + 2, 1};
    jobs.j[2] = jobs.j[1];
    unsigned* scans_done = (unsigned*)(ws + L.scans_done);
    size_t center_lds = (size_t)L.nsb * 12 > (size_t)L.nt_cells * 4 ? (size_t)L.nsb * 12 : (size_t)L.nt_cells * 4;
    if ((size_t)L.nkeys * 4 + 16 > center_lds) center_lds = (size_t)L.nkeys * 4 + 16;     // pull route: the key starts
    // the order job: 1024 class counters + a class per tile (2 B) + room to stage the forward list (2 B per item)
    const size_t n_fwd_items = (size_t)L.ntb + L.ntb / 8 + 8;
    const size_t order_lds_staged = 4096 + ((size_t)L.ntb + 2) * 2 + n_fwd_items * 2 + 16;
    const int staged = tile_order && order_lds_staged <= 150 * 1024;
    const size_t order_lds = staged ? order_lds_staged : 4096 + ((size_t)L.ntb + 2) * 2 + 16;
    if (tile_order && center_lds < order_lds) center_lds = order_lds;
    hipLaunchKernelGGL(center_scatter_kernel, dim3(grid + 3), dim3(BIN_THREADS), center_lds, s,
                       (const int32_t*)tiles_per_gauss, means2d, radii, CN, L.g, (float)tile_size, L.nsb,
                       depths, (const unsigned*)chist, ccursor, sorted, cmeta, (int32_t*)(ws + L.cstart), jobs, scans_done, meta_dev,
                       meta_mirror, seq, (const unsigned*)whint, (const unsigned*)wstat, tile_order, L.ntb, staged,
                       g_sc_raster_split, g_sc_raster_bwd_split ? 0 : 1 SC_DIAG_ARG(g_sc_debug[1] >> 16));
    SC_LAUNCH_CHECK();
    return SC_OK;
}

// Re-zeroes the sort phase's bucket cursors and fallback flags.  Not needed in the normal flow (one
// memset per frame in sc_isect_bin_count covers them and a launch with too small capacities returns
// before touching them); the wrapper calls it before any SECOND sc_isect_bin_sort of the same count
// phase so that a repeated scatter can never start from advanced cursors.
extern "C" int sc_isect_bin_reset_cursors(void* count_workspace, int64_t CN, int C, int tile_width, int tile_height,
                                          sc_stream_t stream) {
    if (!count_workspace || C <= 0 || tile_width <= 0 || tile_height <= 0) return SC_EINVAL;
    if ((int64_t)C * tile_width * tile_height > BIN_MAX_TILES) return SC_EUNSUPPORTED;
    const BinLayout L = bin_layout(CN, C, 1, tile_width, tile_height);
    SC_HIP(hipMemsetAsync((unsigned char*)count_workspace + L.rcursor, 0, L.scans_done - L.rcursor, sc_s(stream)));   // rcursor + nseg
    return SC_OK;
}

extern "C" int sc_isect_ids_rebuild(const int32_t* flatten_ids, const int32_t* isect_offsets, const float* depths,
                                    int C, int N, int tile_width, int tile_height, int64_t n_isects,
                                    int64_t* isect_ids, sc_stream_t stream) {
    if (C < 0 || N < 0 || tile_width <= 0 || tile_height <= 0 || n_isects < 0 || n_isects > 0x7fffffffLL) return SC_EINVAL;
    if (C == 0 || n_isects == 0) return SC_OK;
    if (!flatten_ids || !isect_offsets || !depths || !isect_ids) return SC_EINVAL;
    const int64_t nb = (int64_t)C * tile_width * tile_height;
    if (nb > 0x7fffffffLL) return SC_EINVAL;
    hipLaunchKernelGGL(isect_ids_rebuild_kernel, dim3((unsigned)nb), dim3(256), 0, sc_s(stream), flatten_ids, isect_offsets,
                       depths, (int)nb, tile_width * tile_height, sc_bits_for((int64_t)tile_width * tile_height),
                       (int64_t)C * N, n_isects, isect_ids);
    SC_LAUNCH_CHECK();
    return SC_OK;
}

extern "C" int sc_isect_bin_sort(const float* means2d, const int32_t* radii, const float* depths, int C,
                                 int N, int tile_size, int tile_width, int tile_height,
                                 const int32_t* isect_offsets, const int64_t* meta_dev,
                                 void* count_workspace, int64_t capacity, int64_t rec_capacity,
                                 int64_t super_capacity, int64_t* isect_ids, int32_t* flatten_ids,
                                 void* workspace, size_t ws_bytes, sc_stream_t stream) {
    if (C < 0 || N < 0 || tile_size <= 0 || tile_width <= 0 || tile_height <= 0 || capacity < 0 ||
        rec_capacity < 0 || super_capacity < 0)
        return SC_EINVAL;
    const int64_t CN = (int64_t)C * N;
    const int64_t nb64 = (int64_t)C * tile_width * tile_height;
    if (nb64 > BIN_MAX_TILES || CN >= (1LL << 28) || capacity > 0x7fffffffLL) return SC_EUNSUPPORTED;
    if (capacity == 0 || CN == 0) return SC_OK;
    // Buckets of up to SS_MAX_CAP records are sorted by one workgroup in LDS; when the caller provisions for
    // larger ones (super_capacity > SS_MAX_CAP) they are first cut into depth ranges (big_split_kernel).
    if (super_capacity > BIG_MAX_SUPER) return SC_EUNSUPPORTED;
    const bool big = super_capacity > SS_MAX_CAP;
    int cap = big ? SS_MAX_CAP : (int)((super_capacity + 255) / 256 * 256);
    if (cap < 256) cap = 256;
    if (!means2d || !radii || !depths || !isect_offsets || !meta_dev || !count_workspace || !flatten_ids ||
        !workspace)
        return SC_EINVAL;
    if (ws_bytes < sc_isect_bin_workspace_bytes(CN, C, tile_width, tile_height, rec_capacity)) return SC_EWORKSPACE;
    hipStream_t s = sc_s(stream);
    const BinLayout L = bin_layout(CN, C, N, tile_width, tile_height);
    unsigned char* cws = (unsigned char*)count_workspace;
    const int32_t* soffsets = (const int32_t*)(cws + L.soffsets);
    const int64_t* cmeta = (const int64_t*)(cws + L.cmeta);
    const uint4* sorted = (const uint4*)(cws + L.sorted);
    // zero since sc_isect_bin_count; consumed by the one launch whose capacities pass the device-side
    // check (kernels of a launch with too small capacities return before touching them)
    unsigned* cursor = (unsigned*)(cws + L.rcursor);
    unsigned* n_segs = (unsigned*)(cws + L.nseg);
    const size_t rec_bytes = sc_align_up((size_t)rec_capacity * 8, 256);
    uint2* records = (uint2*)workspace;
    uint2* temp = (uint2*)((unsigned char*)workspace + rec_bytes);
    Segment* segs = (Segment*)((unsigned char*)workspace + 2 * rec_bytes);
    const int seg_bound = big ? (int)seg_bound_for(rec_capacity, L.nsb) : 0;
    const bool pull = L.g.ncls > 0;
    if (pull && cap < SS_SMALL_CAP) cap = SS_SMALL_CAP;      // (the gather's candidate table lives in 6 B x cap of the sort's LDS)
    const int32_t* cstart = (const int32_t*)(cws + L.cstart);
    // 64 Gaussians per wave, 512 per workgroup -- unless the set is a few (<= 262 144) HUGE splats (a sky: more than 32
    // records per Gaussian, walked rectangle by rectangle by whole waves) or tiny: then 16 per wave, so that every SIMD gets
    // a wave.  (Round 2 chose by the Gaussian count alone: S-100k, 9 records per Gaussian, took the sky's variant and
    // 2360 workgroups of fixed table work, 23 us.)
    const bool few_huge = CN <= 262144 && (rec_capacity > 32 * CN || CN < 32768);
    if (pull) {
        // PULL route: no scatter launch and no records buffer -- every bucket's sort workgroup gathers its own records;
        // only oversized buckets (when provisioned for) are gathered into their slice of `records` for big_split_kernel
        if (big) {
            SC_HIP(bin_attrs_once());
            hipLaunchKernelGGL(big_pull_kernel, dim3((unsigned)L.nsb), dim3(BP_THREADS),
                               (BP_CAND + pull_lds_words(L.pt.nrows)) * 4, s, sorted, cstart, cmeta, L.pt, L.g, soffsets, L.nsb,
                               meta_dev, capacity, rec_capacity, super_capacity, cap, records);
        }
    } else if (!few_huge)
        hipLaunchKernelGGL((bin_scatter_flat_kernel<FLAT_THREADS, 64>), dim3((unsigned)((CN + FLAT_THREADS - 1) / FLAT_THREADS)),
                           dim3(FLAT_THREADS), (size_t)L.nsb * 8 + 16 + (FLAT_THREADS / 64) * sizeof(FlatTab), s, sorted, cmeta,
                           L.g, L.nsb, soffsets, meta_dev, capacity, rec_capacity, super_capacity, cursor, records
                           SC_DIAG_ARG(g_sc_debug[0]));
    else {
        constexpr int gpb = FLAT_THREADS_SMALL / 64 * 16;
        hipLaunchKernelGGL((bin_scatter_flat_kernel<FLAT_THREADS_SMALL, 16>), dim3((unsigned)((CN + gpb - 1) / gpb)),
                           dim3(FLAT_THREADS_SMALL), (size_t)L.nsb * 8 + 16 + (FLAT_THREADS_SMALL / 64) * sizeof(FlatTab), s,
                           sorted, cmeta, L.g, L.nsb, soffsets, meta_dev, capacity, rec_capacity, super_capacity, cursor,
                           records SC_DIAG_ARG(g_sc_debug[0]));
    }
    SC_LAUNCH_CHECK();
    const int tile_bits = sc_bits_for(L.g.T);
    SC_HIP(bin_attrs_once());
    if (big) {
        const int bgrid = L.nsb;          // one workgroup per super-tile: the hardware balances the oversized ones
        hipLaunchKernelGGL(big_split_kernel, dim3(bgrid), dim3(BS_THREADS), 0, s, (const uint2*)records, temp, segs,
                           n_segs, seg_bound, soffsets, L.nsb, meta_dev, capacity, rec_capacity, super_capacity, cap SC_DIAG_ARG(g_sc_debug[3]));
        SC_LAUNCH_CHECK();
    }
    const unsigned nsb8 = (unsigned)((L.nsb + 7) / 8 * 8);     // (pull: the XCD-banded bucket map needs whole rounds of 8 blocks)
#define SC_LAUNCH_SORT(THREADSP, RPTP, PULLP, GRID, LDSB, SEGB)                                                               \
    hipLaunchKernelGGL((super_sort_kernel<THREADSP, RPTP, PULLP>), dim3(GRID), dim3(THREADSP), LDSB, s,                       \
                       (const uint2*)records, records, (const uint2*)temp, (const Segment*)segs, (const unsigned*)n_segs, SEGB, \
                       soffsets, L.nsb, L.ntb, L.g, isect_offsets, meta_dev, capacity, rec_capacity, super_capacity, tile_bits,  \
                       cap, isect_ids, flatten_ids, sorted, cstart, cmeta, L.pt SC_DIAG_ARG(g_sc_debug[2]))
    if (!big && cap <= SS_SMALL_CAP) {
        if (pull) SC_LAUNCH_SORT(SS_SMALL_THREADS, SS_SMALL_RPT, true, nsb8, sort_lds_bytes(cap, SS_SMALL_THREADS), 0);
        else SC_LAUNCH_SORT(SS_SMALL_THREADS, SS_SMALL_RPT, false, (unsigned)L.nsb, sort_lds_bytes(cap, SS_SMALL_THREADS), 0);
    } else {
        if (pull) SC_LAUNCH_SORT(SS_THREADS, SS_RPT, true, (unsigned)seg_bound + nsb8, sort_lds_bytes(cap), seg_bound);
        else SC_LAUNCH_SORT(SS_THREADS, SS_RPT, false, (unsigned)(seg_bound + L.nsb), sort_lds_bytes(cap), seg_bound);
    }
#undef SC_LAUNCH_SORT
    SC_LAUNCH_CHECK();
    return SC_OK;
}
