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
constexpr int CNT_GPT = 8;                       // ... and in the count pass, whose cost is the flush of the
constexpr int CNT_GPB = BIN_THREADS * CNT_GPT;   // per-workgroup grids (A/B on S-1M: 2 -> 36 us, 4 -> 23, 8 -> 19, 16 -> 25)
constexpr int BIN_MAX_TILES = 16384;             // C * tile_width * tile_height handled by this path
constexpr int BIN_BIG = 32;                      // rectangles larger than this are walked by a whole wave
constexpr unsigned ID_MASK = 0x0fffffffu;        // flat id lives in the low 28 bits of a record
constexpr unsigned long long KEY_MASK = 0xffffffff0fffffffull;   // (depth, id) without the tile mask

struct Rect { int x0, x1, y0, y1; };

struct Geo {
    int tile_width, tile_height, T;      // tiles
    int ss;                              // super-tile shift (1: 2x2 tiles, 0: super-tile == tile)
    int stw, sth, ST;                    // super-tiles
    int N;
};

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

// ---- pass 1: counts ---------------------------------------------------------------------------
// dgrid_t : [C][th+1][tw+1]   2-D difference grid over tiles        (-> per-tile counts)
// dgrid_s : [C][sth+1][stw+1] 2-D difference grid over super-tiles  (-> records per super-tile)
// chist   : [C][sth][stw]     visible Gaussians by centre super-tile
__global__ __launch_bounds__(BIN_THREADS) void bin_count_kernel(
    const float* __restrict__ means2d, const int32_t* __restrict__ radii, int64_t CN, Geo g,
    float tile_size, int C, int32_t* __restrict__ tiles_per_gauss, int* __restrict__ dgrid_t,
    int* __restrict__ dgrid_s, unsigned* __restrict__ chist) {
    extern __shared__ int lds_i[];
    const int nt = C * (g.tile_height + 1) * (g.tile_width + 1);
    const int ns = g.ss ? C * (g.sth + 1) * (g.stw + 1) : 0;
    const int nc = C * g.ST;
    int* dt = lds_i;
    int* ds = lds_i + nt;
    int* ch = lds_i + nt + ns;
    for (int i = threadIdx.x; i < nt + ns + nc; i += BIN_THREADS) lds_i[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * CNT_GPB;
#pragma unroll
    for (int k = 0; k < CNT_GPT; ++k) {
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
        const int cx = (s.x0 + s.x1 - 1) >> 1, cy = (s.y0 + s.y1 - 1) >> 1;
        atomicAdd(&ch[cam * g.ST + cy * g.stw + cx], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nt; i += BIN_THREADS) { const int v = dt[i]; if (v) atomicAdd(&dgrid_t[i], v); }
    for (int i = threadIdx.x; i < ns; i += BIN_THREADS) { const int v = ds[i]; if (v) atomicAdd(&dgrid_s[i], v); }
    for (int i = threadIdx.x; i < nc; i += BIN_THREADS) { const int v = ch[i]; if (v) atomicAdd(&chist[i], (unsigned)v); }
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
    uint4* __restrict__ sorted, int64_t* __restrict__ cmeta, ScanJobs jobs,
    unsigned* __restrict__ scans_done, int64_t* __restrict__ meta_dev, int64_t* meta_mirror, int64_t seq) {
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
    const int cblock = (int)blockIdx.x - 2;
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
constexpr int FLAT_WAVES = FLAT_THREADS / 64;
struct FlatTab {                 // per wave
    int rx0[64], rx1[64], ry0[64], ry1[64];        // tile rectangle (for the 2x2 tile mask)
    int sx0[64], sy0[64], sw[64], inv[64];         // super-tile rectangle origin, width, 65536 / width + 1
    int base[64], off[64];                         // camera bucket base, first pair number
    unsigned depth[64], id[64];
    unsigned char owner[64 * BIN_BIG];             // pair number -> lane
};

__global__ __launch_bounds__(FLAT_THREADS) void bin_scatter_flat_kernel(
    const uint4* __restrict__ sorted, const int64_t* __restrict__ n_visible, Geo g, int n_sbuckets,
    const int32_t* __restrict__ soffsets, const int64_t* __restrict__ meta, int64_t capacity,
    int64_t rec_capacity, int64_t super_capacity, unsigned* __restrict__ cursor,
    uint2* __restrict__ records, int dbg) {
    extern __shared__ unsigned lds[];
    // the caller may have sized the buffers from a prediction: do nothing if they are too small
    if (meta[0] > capacity || meta[2] > rec_capacity || meta[3] > super_capacity) return;
    const int64_t M = n_visible[0];
    const int64_t base_j = (int64_t)blockIdx.x * FLAT_THREADS;
    if (base_j >= M) return;
    unsigned* hist = lds;                  // [n_sbuckets] counts, then running local cursors
    unsigned* gbase = lds + n_sbuckets;    // [n_sbuckets] global start of this workgroup's slice
    const int lane = sc_lane(), wave = threadIdx.x >> 6;
    FlatTab& tab = reinterpret_cast<FlatTab*>(lds + 2 * n_sbuckets)[wave];
    for (int b = threadIdx.x; b < n_sbuckets; b += FLAT_THREADS) hist[b] = 0;

    // consecutive lanes take consecutive Gaussians of the spatial order
    const int64_t j = base_j + threadIdx.x;
    Rect r = {0, 0, 0, 0}, sr = {0, 0, 0, 0};
    int cam_base = 0;
    unsigned depth = 0, id = 0;
    if (j < M) {
        const uint4 pay = sorted[j];
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
    tab.rx0[lane] = r.x0; tab.rx1[lane] = r.x1; tab.ry0[lane] = r.y0; tab.ry1[lane] = r.y1;
    tab.sx0[lane] = sr.x0; tab.sy0[lane] = sr.y0; tab.sw[lane] = sw;
    tab.inv[lane] = sw > 0 ? 65536 / sw + 1 : 0;              // q / sw == (q * inv) >> 16 for q < 1024
    tab.base[lane] = cam_base; tab.off[lane] = off; tab.depth[lane] = depth; tab.id[lane] = id;
    for (int q = 0; q < c; ++q) tab.owner[off + q] = (unsigned char)lane;
    __syncthreads();                       // hist zeroed, tables complete

    // decode pair number p -> bucket (and, for pass 2, everything the record needs)
    auto bucket_of = [&](int p, int& o, int& sx, int& sy) -> int {
        o = tab.owner[p];
        const int q = p - tab.off[o];
        const int row = (q * tab.inv[o]) >> 16;
        sy = tab.sy0[o] + row;
        sx = tab.sx0[o] + (q - row * tab.sw[o]);
        return tab.base[o] + sy * g.stw + sx;
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
            for (int q = lane; q < bcnt; q += 64) {
                const int sy = by0 + q / bw, sx = bx0 + q % bw;
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
    for (int b = threadIdx.x; b < n_sbuckets; b += FLAT_THREADS) {
        const unsigned cb = hist[b];
        if (cb) gbase[b] = (unsigned)soffsets[b] + atomicAdd(&cursor[b], cb);
        hist[b] = 0;
    }
    __syncthreads();
    if (dbg & 2) return;
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
        const unsigned mask = tile_mask(tab.rx0[o], tab.rx1[o], tab.ry0[o], tab.ry1[o], sx, sy);
        if (!(dbg & 1)) records[slot] = make_uint2(tab.depth[o], tab.id[o] | (mask << 28));
    }
    walk_big([&](int b, const Rect& br, int sx, int sy, unsigned bd, unsigned bi) {
        const unsigned slot = gbase[b] + atomicAdd(&hist[b], 1u);
        if (!(dbg & 1)) records[slot] = make_uint2(bd, bi | (tile_mask(br.x0, br.x1, br.y0, br.y1, sx, sy) << 28));
    });
}

// ---- pass 3: per-super-tile sort + emit ---------------------------------------------------------------
constexpr int SS_THREADS = 512;
constexpr int SS_WAVES = SS_THREADS / 64;
constexpr int SS_MAX_CAP = 7168;                 // records of the largest super-tile this path sorts in LDS
constexpr int SS_RPT = SS_MAX_CAP / SS_THREADS;  // records per thread held in registers (14)

// Emits the per-tile lists of one super-tile from its records sorted on (depth, id) in LDS.
// Stable filter per tile: rank of record i in tile k's list = number of records j < i with mask bit k;
// computed per (round, wave) with ballots, bases by a tiny scan over the (round, wave) table.
// `order` (nullable): when given, the sorted sequence is S[order[0]], S[order[1]], ... (the
// interpolation sort keeps 2-byte ranks instead of a second copy of the keys: 14 instead of 20 B of LDS
// per record, i.e. three resident workgroups per CU instead of two)
template <int THREADS>
__device__ __forceinline__ void emit_tiles(const unsigned long long* __restrict__ S,
                                           const unsigned short* __restrict__ order, int n, int sb, const Geo& g,
                                           const int32_t* __restrict__ offsets, int n_tiles_total,
                                           int64_t n_isects, int tile_bits, unsigned long long* table,
                                           int64_t* __restrict__ isect_ids, int32_t* __restrict__ flatten_ids,
                                           int dbg) {
    constexpr int WAVES = THREADS / 64;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cam = sb / g.ST, srem = sb - cam * g.ST;
    const int sy = srem / g.stw, sx = srem - sy * g.stw;
    const int ntile = g.ss ? 4 : 1;
    const int rounds = (n + THREADS - 1) / THREADS;
    // pass A: per (round, wave) packed counts (16 bits per tile; a tile list has <= 9216 entries)
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
        tbase[k] = tok[k] ? offsets[tflat] : 0;
        // end of this tile's list: a write is dropped rather than allowed past it, so that records
        // whose masks disagree with the counts (corrupt input, diagnostic skips) cannot go out of bounds
        tend[k] = tok[k] ? ((tflat + 1 < n_tiles_total) ? offsets[tflat + 1] : (int)n_isects) : 0;
        hi_key[k] = ((long long)cam << (32 + tile_bits)) | ((long long)tile << 32);
    }
    if (dbg & 1) return;
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
                    if (isect_ids && !(dbg & 16)) isect_ids[pos] = hi_key[k] | (long long)(key >> 32);
                    flatten_ids[pos] = (int32_t)((unsigned)key & ID_MASK);
                }
            }
        }
    }
}

// interpolation sort: the common path.  Records are spread over `nbk` sub-buckets by a MONOTONE map
// of the depth bit pattern, grouped by one counting pass, and ranked inside their sub-bucket by
// counting smaller (depth, id) keys.  A super-tile whose keys pile up in one sub-bucket (occupancy
// > BS_MAX_OCC, e.g. hundreds of equal depths) is flagged and left to super_radix_kernel.
constexpr int BS_MAX_OCC = 48;

__global__ __launch_bounds__(SS_THREADS) void super_sort_kernel(
    const uint2* __restrict__ records, const int32_t* __restrict__ soffsets, int n_sbuckets, int n_tbuckets,
    Geo g, const int32_t* __restrict__ offsets, const int64_t* __restrict__ meta, int64_t capacity,
    int64_t rec_capacity, int64_t super_capacity, int tile_bits, int cap, int per_thread,
    unsigned char* __restrict__ needs_radix, int64_t* __restrict__ isect_ids, int32_t* __restrict__ flatten_ids,
    int dbg) {
    extern __shared__ __align__(16) unsigned char smem[];
    // [B: cap u64][order: cap u16][boff: SS_THREADS*per_thread + 1 u32][table]
    unsigned long long* B = reinterpret_cast<unsigned long long*>(smem);
    unsigned short* order = reinterpret_cast<unsigned short*>(B + cap);      // cap is a multiple of 256
    unsigned* boff = reinterpret_cast<unsigned*>(order + cap);
    const int nbk = SS_THREADS * per_thread;
    unsigned long long* table = reinterpret_cast<unsigned long long*>(boff + ((nbk + 2) & ~1));
    __shared__ unsigned red_lo[SS_WAVES], red_hi[SS_WAVES], red_sum[SS_WAVES], red_occ[SS_WAVES];

    // the SAME three comparisons in every kernel of the sort phase and in the host wrapper
    // (rendering._bin_launch_ran): a launch either runs in full or not at all
    if (meta[0] > capacity || meta[2] > rec_capacity || meta[3] > super_capacity) return;
    const int sb = blockIdx.x;
    const int s = soffsets[sb];
    const int e = (sb + 1 < n_sbuckets) ? soffsets[sb + 1] : (int)meta[2];
    const int n = e - s;
    if (n <= 0) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    for (int i = t; i <= nbk; i += SS_THREADS) boff[i] = 0;
    // the super-tile's records are read from global memory ONCE, all loads of a thread in flight
    // together, and stay in registers for the min/max, counting and scatter passes
    // (cap <= SS_MAX_CAP -> at most SS_RPT records per thread).  Phase times on S-1M (rocprof A/B):
    // loads + min/max 17 us, counting +5, scan + scatter +10, ranking +24, emit +28.
    uint2 rec[SS_RPT];
#pragma unroll
    for (int k = 0; k < SS_RPT; ++k) {
        const int i = t + k * SS_THREADS;
        rec[k] = (i < n) ? records[s + i] : make_uint2(0u, 0u);
    }
    unsigned lo = 0xffffffffu, hi = 0u;
#pragma unroll
    for (int k = 0; k < SS_RPT; ++k) {
        if (t + k * SS_THREADS < n) { lo = min(lo, rec[k].x); hi = max(hi, rec[k].x); }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, o, 64));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, o, 64));
    }
    if (lane == 0) { red_lo[wave] = lo; red_hi[wave] = hi; }
    __syncthreads();
    lo = red_lo[0]; hi = red_hi[0];
#pragma unroll
    for (int w = 1; w < SS_WAVES; ++w) { lo = min(lo, red_lo[w]); hi = max(hi, red_hi[w]); }
    const float scale = (float)nbk / ((float)(hi - lo) + 1.0f);
    auto sub_bucket = [&](unsigned d) -> int {
        const int v = (int)((float)(d - lo) * scale);     // monotone in d
        return min(v, nbk - 1);
    };
#pragma unroll
    for (int k = 0; k < SS_RPT; ++k)
        if (t + k * SS_THREADS < n) atomicAdd(&boff[sub_bucket(rec[k].x)], 1u);
    __syncthreads();
    // exclusive scan: thread t owns per_thread consecutive counters (per_thread is ODD: the
    // lanes' strides then hit 32 distinct banks instead of two)
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
    occ = red_occ[0];
#pragma unroll
    for (int w = 1; w < SS_WAVES; ++w) occ = max(occ, red_occ[w]);
    if (occ > BS_MAX_OCC) {                     // workgroup-uniform
        if (t == 0) needs_radix[sb] = 1;
        return;
    }
    unsigned run = incl - sum;
#pragma unroll
    for (int w = 0; w < SS_WAVES; ++w) if (w < wave) run += red_sum[w];
    for (int k = 0; k < per_thread; ++k) {      // counts -> sub-bucket starts (used as running cursors)
        const unsigned c = boff[t * per_thread + k];
        boff[t * per_thread + k] = run;
        run += c;
    }
    __syncthreads();
    // scatter: after this pass boff[j] is the END of sub-bucket j (== start of j+1)
#pragma unroll
    for (int k = 0; k < SS_RPT; ++k) {
        if (t + k * SS_THREADS < n) {
            const uint2 r = rec[k];
            const unsigned slot = atomicAdd(&boff[sub_bucket(r.x)], 1u);
            B[slot] = ((unsigned long long)r.x << 32) | r.y;
        }
    }
    __syncthreads();
    if (dbg & 2) return;
    // rank inside the sub-bucket by (depth bits, flat id) -> order[rank] = position in B
    for (int p = t; p < n; p += SS_THREADS) {
        const unsigned long long key = B[p];
        const unsigned long long kk = key & KEY_MASK;
        const int j = sub_bucket((unsigned)(key >> 32));
        const unsigned beg = j > 0 ? boff[j - 1] : 0u, end = boff[j];
        unsigned r = beg;
        for (unsigned q = beg; q < end; ++q) r += ((B[q] & KEY_MASK) < kk) ? 1u : 0u;
        order[r] = (unsigned short)p;
    }
    __syncthreads();
    emit_tiles<SS_THREADS>(B, order, n, sb, g, offsets, n_tbuckets, meta[0], tile_bits, table, isect_ids, flatten_ids, dbg);
}

// ---- radix fallback for flagged super-tiles --------------------------------------------------------------
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

// Persistent-style grid: every workgroup strides over the super-tiles and sorts only the flagged
// ones (normally none: the launch then costs a few hundred flag reads).
__global__ __launch_bounds__(TS_THREADS) void super_radix_kernel(
    const uint2* __restrict__ records, const int32_t* __restrict__ soffsets, int n_sbuckets, int n_tbuckets,
    Geo g, const int32_t* __restrict__ offsets, const int64_t* __restrict__ meta, int64_t capacity,
    int64_t rec_capacity, int64_t super_capacity, int tile_bits, int id_bits, int cap,
    const unsigned char* __restrict__ needs_radix, int64_t* __restrict__ isect_ids,
    int32_t* __restrict__ flatten_ids) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned long long* A = reinterpret_cast<unsigned long long*>(smem);
    unsigned long long* Bb = A + cap;
    unsigned long long* table = Bb + cap;
    __shared__ unsigned h[TS_WAVES][256];
    __shared__ unsigned wtot[TS_WAVES];
    __shared__ unsigned diff_s;
    if (meta[0] > capacity || meta[2] > rec_capacity || meta[3] > super_capacity) return;
    const int t = threadIdx.x;
    for (int sb = blockIdx.x; sb < n_sbuckets; sb += gridDim.x) {
        if (!needs_radix[sb]) continue;              // workgroup-uniform
        const int s = soffsets[sb];
        const int e = (sb + 1 < n_sbuckets) ? soffsets[sb + 1] : (int)meta[2];
        const int n = e - s;
        if (n <= 0) continue;
        __syncthreads();
        if (t == 0) diff_s = 0;
        __syncthreads();
        const unsigned first_depth = records[s].x;
        unsigned diff = 0;
        for (int i = t; i < n; i += TS_THREADS) {
            const uint2 r = records[s + i];
            A[i] = ((unsigned long long)r.x << 32) | r.y;
            diff |= r.x ^ first_depth;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) diff |= (unsigned)__shfl_xor((int)diff, o, 64);
        if ((t & 63) == 0 && diff) atomicOr(&diff_s, diff);
        __syncthreads();
        const int sig = 32 - __clz((int)diff_s);      // significant depth bits (0 when all equal)
        int chunk = (n + TS_WAVES - 1) / TS_WAVES;
        chunk = (chunk + 63) & ~63;
        unsigned long long* src = A;
        unsigned long long* dst = Bb;
        // LSD on the full (depth, id) key: id bits first (bits 28..31 hold the tile mask and are
        // skipped), then the significant depth bits
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
        emit_tiles<TS_THREADS>(src, nullptr, n, sb, g, offsets, n_tbuckets, meta[0], tile_bits, table, isect_ids, flatten_ids, 0);
    }
}

}  // namespace

// ---- host side --------------------------------------------------------------------------------
// count-phase workspace (handed to BOTH calls):
//   dgrid_t | dgrid_s | chist | ccursor | rcursor | rflags | scans_done | soffsets | cstart | smeta[2] | cmeta[2] | sorted uint4[CN]
//   (everything before soffsets is zeroed by the ONE memset of a frame; rcursor / rflags are the
//    scatter's bucket cursors and the "needs the radix fallback" flags of the sort phase, kept here so
//    that the sort phase needs no memset of its own: a second memset cost 5 us + a 6 us bubble)
// sort-phase workspace:
//   records uint2[rec_capacity]
struct BinLayout {
    Geo g;
    int C, nt_cells, ns_cells, nsb, ntb;
    size_t dgrid_t, dgrid_s, chist, ccursor, rcursor, rflags, scans_done, soffsets, cstart, smeta, cmeta, sorted, total;
};

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
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += sc_align_up(bytes, 256); return at; };
    L.dgrid_t = take((size_t)L.nt_cells * 4);
    L.dgrid_s = take((size_t)L.ns_cells * 4);
    L.chist = take((size_t)L.nsb * 4);
    L.ccursor = take((size_t)L.nsb * 4);
    L.rcursor = take((size_t)L.nsb * 4);
    L.rflags = take((size_t)L.nsb);
    L.scans_done = take(4);
    L.soffsets = take((size_t)L.nsb * 4);
    L.cstart = take((size_t)L.nsb * 4);
    L.smeta = take(16);
    L.cmeta = take(16);
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
    if ((e = hipFuncSetAttribute((const void*)bin_count_kernel, (hipFuncAttribute)a, 152 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)center_scatter_kernel, (hipFuncAttribute)a, 152 * 1024)) != hipSuccess) return e;
    // (super_sort_kernel also has ~4 KiB of static LDS: stay below 160 KiB in total)
    if ((e = hipFuncSetAttribute((const void*)super_sort_kernel, (hipFuncAttribute)a, 156 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)super_radix_kernel, (hipFuncAttribute)a, 150 * 1024)) != hipSuccess) return e;
    done[dev] = true;
    return hipSuccess;
}

static inline size_t count_lds_bytes(const BinLayout& L) {
    return (size_t)(L.nt_cells + L.ns_cells + L.nsb) * 4;
}

extern "C" size_t sc_isect_bin_workspace_bytes(int64_t CN, int C, int tile_width, int tile_height,
                                               int64_t n_isects) {
    const int64_t nb = (int64_t)C * tile_width * tile_height;
    if (nb <= 0 || nb > BIN_MAX_TILES) return 256;
    const BinLayout L = bin_layout(CN, C, 1, tile_width, tile_height);
    if (n_isects < 0) return L.total;                                  // count-phase workspace
    return sc_align_up((size_t)n_isects * 8, 256) + 256;
}

extern "C" int sc_isect_bin_count(const float* means2d, const int32_t* radii, const float* depths, int C, int N,
                                  int tile_size,
                                  int tile_width, int tile_height, int32_t* tiles_per_gauss,
                                  int32_t* isect_offsets, int64_t* meta_dev, int64_t* meta_mirror,
                                  int64_t seq, void* count_workspace, size_t ws_bytes, sc_stream_t stream) {
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
        return SC_OK;
    }
    if (!means2d || !radii || !depths || !tiles_per_gauss || !isect_offsets || !count_workspace) return SC_EINVAL;
    const BinLayout L = bin_layout(CN, C, N, tile_width, tile_height);
    if (ws_bytes < L.total) return SC_EWORKSPACE;
    if (count_lds_bytes(L) > 150 * 1024 || (size_t)L.nt_cells * 4 > 150 * 1024) return SC_EUNSUPPORTED;
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
    hipLaunchKernelGGL(bin_count_kernel, dim3((unsigned)((CN + CNT_GPB - 1) / CNT_GPB)), dim3(BIN_THREADS),
                       count_lds_bytes(L), s, means2d, radii, CN,
                       L.g, (float)tile_size, C, tiles_per_gauss, dgrid_t, dgrid_s, chist);
    SC_LAUNCH_CHECK();
    // tile grid -> isect_offsets + meta[0..1]; super-tile grid -> record offsets + meta[2..3]: two extra
    // blocks of the centre-scatter launch (see the kernel)
    ScanJobs jobs;
    jobs.j[0] = ScanJob{(const int*)dgrid_t, C, tile_width, tile_height, isect_offsets, meta_dev, 1};
    jobs.j[1] = ScanJob{(const int*)dgrid_s, C, L.g.stw, L.g.sth, soffsets, meta_dev + 2, 1};
    jobs.j[2] = jobs.j[1];
    unsigned* scans_done = (unsigned*)(ws + L.scans_done);
    const size_t center_lds = (size_t)L.nsb * 12 > (size_t)L.nt_cells * 4 ? (size_t)L.nsb * 12 : (size_t)L.nt_cells * 4;
    hipLaunchKernelGGL(center_scatter_kernel, dim3(grid + 2), dim3(BIN_THREADS), center_lds, s,
                       (const int32_t*)tiles_per_gauss, means2d, radii, CN, L.g, (float)tile_size, L.nsb,
                       depths, (const unsigned*)chist, ccursor, sorted, cmeta, jobs, scans_done, meta_dev,
                       meta_mirror, seq);
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
    SC_HIP(hipMemsetAsync((unsigned char*)count_workspace + L.rcursor, 0, L.scans_done - L.rcursor, sc_s(stream)));
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
    // LDS of one super-tile workgroup: 20 B per record (+ table) must fit ~150 KiB
    if (super_capacity > SS_MAX_CAP) return SC_EUNSUPPORTED;
    int cap = (int)((super_capacity + 255) / 256 * 256);
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
    unsigned char* needs_radix = cws + L.rflags;
    uint2* records = (uint2*)workspace;
    hipLaunchKernelGGL(bin_scatter_flat_kernel, dim3((unsigned)((CN + FLAT_THREADS - 1) / FLAT_THREADS)),
                       dim3(FLAT_THREADS), (size_t)L.nsb * 8 + FLAT_WAVES * sizeof(FlatTab), s, sorted, cmeta, L.g, L.nsb,
                       soffsets, meta_dev, capacity, rec_capacity, super_capacity, cursor, records, g_sc_debug[0]);
    SC_LAUNCH_CHECK();
    const int tile_bits = sc_bits_for(L.g.T);
    const int id_bits = sc_bits_for(CN > 1 ? CN - 1 : 1);
    // sub-buckets per thread: ~1 per record (keeps two workgroups resident per CU), rounded up to
    // an ODD count (bank-conflict-free scan)
    int per_thread = (cap + SS_THREADS - 1) / SS_THREADS;
    per_thread |= 1;
    const int nbk = SS_THREADS * per_thread;
    const size_t table_bytes = (size_t)((cap + SS_THREADS - 1) / SS_THREADS + 1) * SS_WAVES * 8 + 64;
    const size_t lds_sort = (size_t)cap * 10 + (size_t)((nbk + 2) & ~1) * 4 + table_bytes;
    const size_t table_radix = (size_t)((cap + TS_THREADS - 1) / TS_THREADS + 1) * TS_WAVES * 8 + 64;
    const size_t lds_radix = (size_t)cap * 16 + table_radix;
    if (lds_sort > 156 * 1024) return SC_EUNSUPPORTED;
    SC_HIP(bin_attrs_once());
    hipLaunchKernelGGL(super_sort_kernel, dim3(L.nsb), dim3(SS_THREADS), lds_sort, s, (const uint2*)records, soffsets,
                       L.nsb, L.ntb, L.g, isect_offsets, meta_dev, capacity, rec_capacity, super_capacity, tile_bits, cap,
                       per_thread, needs_radix, isect_ids, flatten_ids, g_sc_debug[2]);
    SC_LAUNCH_CHECK();
    const int rgrid = L.nsb < 512 ? L.nsb : 512;      // persistent; A/B when no super-tile is flagged: 128 -> 8.5 us, 512 -> 4.8 us
    hipLaunchKernelGGL(super_radix_kernel, dim3(rgrid), dim3(TS_THREADS), lds_radix, s, (const uint2*)records, soffsets,
                       L.nsb, L.ntb, L.g, isect_offsets, meta_dev, capacity, rec_capacity, super_capacity, tile_bits, id_bits,
                       cap, (const unsigned char*)needs_radix, isect_ids, flatten_ids);
    SC_LAUNCH_CHECK();
    return SC_OK;
}
