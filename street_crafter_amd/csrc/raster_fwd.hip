// a9: rasterize_to_pixels forward for gfx950.
// Replaces gsplat.rendering.rasterize_to_pixels as called at
// street_gaussian/models/street_gaussian_renderer.py:267-280 (semantics: SURVEY.md A.5).
//
// Two kernels, selectable at run time with sc_set_option("raster_fwd", v):
//   0  reference-shaped: one lane per pixel, every lane evaluates every splat of the tile; any
//      tile_size <= 32 and any channel count <= 32.  Generic fallback and on-GPU cross-check.
//   3  (default; tile 16, 3 or 4 channels) one wave per tile with an exact tile-level cull: see
//      raster_fwd_wave_kernel.  Bit-identical to 0 (same pinned arithmetic, raster_common.h).
// (Round 1 also carried a 4-wave culled kernel, its software-pipelined form and a packed-record
//  variant, numbered 1, 2 and 4; all measured slower -- DESIGN.md, "Tried and dropped" -- and removed.)
// The blend itself uses v_exp_f32 (fast exp) and FMA contraction: pixels agree with the oracle
// to ~1e-6 relative, not bitwise (tolerance stated in tests/test_gpu_parity.py).
#include "raster_common.h"

int g_sc_raster_fwd_variant = 3;


namespace {

#ifdef SC_RASTER_SB
constexpr int RASTER_SB = SC_RASTER_SB;     // (experiment builds: tools/ab_lib.py)
#else
constexpr int RASTER_SB = 1;                // splats staged per lane per batch of the wave kernel (see raster_item)
#endif


// ------------------------------------------------------------------------------------------
// variant 0: reference-shaped
// ------------------------------------------------------------------------------------------
template <int CDIM>
__global__ void raster_fwd_ref_kernel(
    const float* __restrict__ means2d, const float* __restrict__ conics,
    const float* __restrict__ colors, const float* __restrict__ opacities,
    const float* __restrict__ backgrounds, const uint8_t* __restrict__ tile_masks, int N, int D,
    int width, int height, int tile_size, int tile_width, int tile_height,
    const int32_t* __restrict__ isect_offsets, const int32_t* __restrict__ flatten_ids, int n_isects,
    float* __restrict__ render_colors, float* __restrict__ render_alphas,
    int32_t* __restrict__ last_ids) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int B = blockDim.x * blockDim.y;
    int* id_s = reinterpret_cast<int*>(smem);                              // [B]
    float4* xyoa_s = reinterpret_cast<float4*>(smem + (size_t)B * 16);     // [B] mx,my,opac,conic.a
    float2* bc_s = reinterpret_cast<float2*>(smem + (size_t)B * 32);       // [B] conic.b, conic.c

    const int cam = blockIdx.z;
    const int tile_id = blockIdx.y * tile_width + blockIdx.x;
    const int tflat = cam * tile_width * tile_height + tile_id;
    const int px_i = blockIdx.x * tile_size + threadIdx.x;
    const int py_i = blockIdx.y * tile_size + threadIdx.y;
    const float px = (float)px_i + 0.5f, py = (float)py_i + 0.5f;
    const bool inside = (px_i < width) && (py_i < height);
    const int64_t pix = ((int64_t)cam * height + py_i) * width + px_i;
    const int tr = threadIdx.y * blockDim.x + threadIdx.x;

    if (tile_masks && !tile_masks[tflat]) {
        if (inside) {
            for (int d = 0; d < D; ++d)
                render_colors[pix * D + d] = backgrounds ? backgrounds[cam * D + d] : 0.f;
            render_alphas[pix] = 0.f;
            if (last_ids) last_ids[pix] = 0;
        }
        return;
    }

    const int total_tiles = gridDim.z * tile_width * tile_height;
    int range_start, range_end;
    sc_tile_range(isect_offsets, tflat, total_tiles, n_isects, range_start, range_end);
    const int num_batches = (range_end - range_start + B - 1) / B;

    float T = 1.0f;
    int cur_idx = 0;
    bool done = !inside;
    float pix_out[CDIM > 0 ? CDIM : SC_MAX_CDIM];
    constexpr int ND = CDIM > 0 ? CDIM : SC_MAX_CDIM;
#pragma unroll
    for (int d = 0; d < ND; ++d) pix_out[d] = 0.f;

    for (int b = 0; b < num_batches; ++b) {
        if (__syncthreads_count(done) >= B) break;
        const int batch_start = range_start + B * b;
        const int idx = batch_start + tr;
        if (idx < range_end) {
            const int g = sc_safe_id(flatten_ids[idx], N);
            if (g >= 0) {
                id_s[tr] = g;
                const float2 xy = *reinterpret_cast<const float2*>(means2d + (int64_t)g * 2);
                const float* cn = conics + (int64_t)g * 3;
                const ScSplat sp = sc_prescale(xy.x, xy.y, cn[0], cn[1], cn[2], opacities[g]);
                xyoa_s[tr] = make_float4(sp.mx, sp.my, sp.lop, sp.A2);
                bc_s[tr] = make_float2(sp.B2, sp.C2);
            } else {                                   // dead entry: alpha = exp2(-inf) = 0 -> skipped
                id_s[tr] = 0;
                xyoa_s[tr] = make_float4(0.f, 0.f, -INFINITY, 0.f);
                bc_s[tr] = make_float2(0.f, 0.f);
            }
        }
        __syncthreads();
        const int bsz = min(B, range_end - batch_start);
        for (int t = 0; (t < bsz) && !done; ++t) {
            const float4 a = xyoa_s[t];
            const float2 bc = bc_s[t];
            const float dx = a.x - px, dy = a.y - py;
            const float sigma = sc_sigma2(a.w, sc_row_b(bc.x, dy), sc_row_q(bc.y, dy), dx);
            const float alpha = sc_alpha2(a.z, sigma);
            if (!sc_valid(sigma, alpha)) continue;
            const float next_T = sc_next_T(T, alpha);
            if (next_T <= SC_T_EPS) { done = true; break; }
            const float vis = __fmul_rn(alpha, T);
            const float* c = colors + (int64_t)id_s[t] * D;
#pragma unroll
            for (int d = 0; d < ND; ++d)
                if (CDIM > 0 || d < D) pix_out[d] = __fmaf_rn(c[d], vis, pix_out[d]);
            cur_idx = batch_start + t;
            T = next_T;
        }
    }
    if (inside) {
        render_alphas[pix] = 1.0f - T;
#pragma unroll
        for (int d = 0; d < ND; ++d)
            if (CDIM > 0 || d < D)
                render_colors[pix * D + d] = backgrounds ? pix_out[d] + T * backgrounds[cam * D + d] : pix_out[d];
        if (last_ids) last_ids[pix] = cur_idx;
    }
}

// ------------------------------------------------------------------------------------------
// variant 3: ONE WAVE PER TILE, four pixels per lane (lane l owns x = 4*(l&3)..+3 of row l>>2 of the
// 16x16 tile).  Measured on the 4-wave form the blend loop was co-bound by the LDS broadcast reads of
// the splat record (3 x ds_read_b128 per splat per wave, 4 waves per tile) and by VALU; with one
// wave per tile each record is read once per tile instead of four times, dy-dependent terms are
// shared by the lane's pixels, there is no workgroup barrier at all (the workgroup IS the wave),
// the early exit is a single wave vote, and up to 32 tiles are resident per CU to hide the gather
// latency.  Same pinned arithmetic -> bit-identical to variant 0.
// The body is templated on the part of a tile one wave covers (NSUB = 1 whole tile, 2 = a half: 16x8,
// 2 pixels per lane, 4 = a quarter); only NSUB = 1 is instantiated: cutting tiles into half-tile waves
// behind a heavy-first work list was measured (tools/exp_raster.py, profiles/r02_raster_policy_ab.txt)
// and costs 30 us on S-1M (staging and LDS reads double) for -60 us on the street scene, whose real
// problem was the block -> tile map (see the kernel).
// ------------------------------------------------------------------------------------------
// TRACK: record last_ids (the sorted index of the last splat each pixel blended), needed only by
// the backward pass; inference launches the variant without it.
// NSUB: 1 = whole tile, 2 = half (8 rows), 4 = quarter (4 rows); `sub` = which one.
// PACKED: `means2d` points to one 48-B record per Gaussian, (x, y, conic a, b | conic c, opacity, colour 0, 1 |
// colour 2, 3, -, -), written by projection_sh_fwd_kernel for the fused forward: one gather line per splat instead of four.
// ED: the "RGB+ED" epilogue (sc_rasterize_fwd_ed; CDIM == 4): channel 3 leaves as depth sum / max(alpha, 1e-10).
#ifdef SC_DIAG
// Diagnostic build only ("debug1" bit 4; tools/exp_raster_phases.py): where the waves of the wave kernel spend their
// lives, in shader-clock cycles summed over all waves: [0] entry to exit, [1] issuing the next batch's gathers,
// [2] cull + compaction, [3] blend loop, [4] waves, [5] batches, [6] records kept by the cull, [7] entry to the first
// batch's parameters having arrived, [8] entry to exit in ticks of the constant 100 MHz clock.  (The wait for a batch's parameters at its top measured 72 cycles: the prefetch
// hides the gathers completely, except the first batch's.)
// One row per wave (tile x half): 90 k atomics on eight shared counters made the kernel four times slower.
constexpr int SC_PHASE_ROWS = 65536;
__device__ unsigned long long g_sc_phase_cycles[SC_PHASE_ROWS][10];
__global__ void phase_sum_kernel(unsigned long long* out8, int reset) {
    __shared__ unsigned long long acc[10];
    if (threadIdx.x < 10) acc[threadIdx.x] = 0;
    __syncthreads();
    for (int r = threadIdx.x; r < SC_PHASE_ROWS; r += blockDim.x)
        for (int k = 0; k < 10; ++k) {
            const unsigned long long v = g_sc_phase_cycles[r][k];
            if (v) atomicAdd(&acc[k], v);
            if (reset) g_sc_phase_cycles[r][k] = 0;
        }
    __syncthreads();
    if (threadIdx.x < 10) out8[threadIdx.x] = acc[threadIdx.x];
}
extern "C" int sc_diag_phase_cycles(unsigned long long* out8, int reset) {
    unsigned long long* d = nullptr;
    hipError_t e = hipMalloc(&d, 80);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(phase_sum_kernel, dim3(1), dim3(1024), 0, 0, d, reset);
    e = hipDeviceSynchronize();
    if (e == hipSuccess && out8) e = hipMemcpy(out8, d, 80, hipMemcpyDeviceToHost);      // (ten words)
    hipFree(d);
    return (int)e;
}
#endif

// PLANAR: render_colors is stored [C][CDIM][H][W] (one plane per channel; the caller hands it out as a permuted [C,H,W,CDIM]
// view) instead of interleaved [C][H][W][CDIM]: the reference caller's glue behind the operator (renderer.py:282-300) slices the
// result channel-wise -- `render_colors[..., :-1]`, `[..., -1:] / alpha`, clamp, `.permute(2, 0, 1)` -- which are strided
// kernels over the interleaved buffer and dense ones over planes (16.0 -> 11.0 and 14.6 -> 10.1 us at 1920x1280,
// tools/exp_glue_layout.py).  A lane owns 4 (or 2) consecutive pixels of a row: one 16-B (8-B) store per channel.
template <int CDIM, bool TRACK, int NSUB, bool PACKED = false, bool ED = false, bool PLANAR = false>
__device__ __forceinline__ void raster_item(
    const float* __restrict__ means2d, const float* __restrict__ conics,
    const float* __restrict__ colors, const float* __restrict__ opacities,
    const float* __restrict__ backgrounds, const uint8_t* __restrict__ tile_masks, int N,
    int width, int height, int tile_width, int tile_height, int total_tiles,
    const int32_t* __restrict__ isect_offsets, const int32_t* __restrict__ flatten_ids, int n_isects,
    float* __restrict__ render_colors, float* __restrict__ render_alphas,
    int32_t* __restrict__ last_ids, int tflat, int sub,
    float4* xyoa_s, float4* bck_s, float4* col_s, int32_t* __restrict__ tile_work SC_DIAG_PARAM(dbg)) {
    // Splats staged per lane per batch.  ONE (batches of 64) since round 3: the kernel's second bound, beside VALU issue,
    // is the rate of its parameter gathers (4 random sectors per list entry: with the blend loop compiled out S-1M's
    // whole lists take 686 us, i.e. ~35 us per million entries walked), and a tile that stops after
    // ~250 entries throws away what it staged beyond that point: on average half a batch plus the batch in flight.
    // Batches of 64 halve that: S-1M 142 -> 134 us, sky 107 -> 100, S-100k 135 -> 132, street scene unchanged
    // (profiles/r03_raster_batch_ab.txt).
    constexpr int SB = RASTER_SB;
    constexpr int B = 64 * SB;            // batch size (half-wave batches of 32 were measured too: +6 / +30 / +7 us on
                                          // S-1M / street / S-100k, profiles/r03_raster_batch_ab.txt)
    constexpr int NP = NSUB == 1 ? 2 : 1; // pixel PAIRS per lane
    constexpr int PPL = NSUB == 1 ? 4 : (NSUB == 2 ? 2 : 1);   // live pixels per lane
    constexpr int ROWS = 16 / NSUB;       // rows of the tile this wave covers

    const int tiles_per_cam = tile_width * tile_height;
    const int cam = tflat / tiles_per_cam;
    const int tile_id = tflat - cam * tiles_per_cam;
    const int tyi = tile_id / tile_width, txi = tile_id - tyi * tile_width;
    const int lane = threadIdx.x;
    // lane -> pixels: 16 / PPL lanes per row
    const int lanes_per_row = 16 / PPL;
    const int row = sub * ROWS + lane / lanes_per_row;
    const int px0_i = txi * 16 + PPL * (lane % lanes_per_row), py_i = tyi * 16 + row;
    const float py = (float)py_i + 0.5f;
    float pxf[2 * NP];
    bool inside[2 * NP];
#pragma unroll
    for (int k = 0; k < 2 * NP; ++k) {
        pxf[k] = (float)(px0_i + k) + 0.5f;
        inside[k] = (k < PPL) && (px0_i + k < width) && (py_i < height);
    }
    const int64_t pix0 = ((int64_t)cam * height + py_i) * width + px0_i;

    // (PLANAR) first pixel of this lane in channel plane 0 of its camera; plane stride = height * width
    const int64_t plane = (int64_t)height * width;
    const int64_t ppix0 = (int64_t)cam * CDIM * plane + (int64_t)py_i * width + px0_i;
    if (tile_masks && !tile_masks[tflat]) {
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            if (inside[k]) {
#pragma unroll
                for (int d = 0; d < CDIM; ++d)
                    render_colors[PLANAR ? ppix0 + d * plane + k : (pix0 + k) * CDIM + d] = backgrounds ? backgrounds[cam * CDIM + d] : 0.f;
                render_alphas[pix0 + k] = 0.f;
                if (last_ids) last_ids[pix0 + k] = 0;
            }
        }
        if (tile_work && lane == 0) tile_work[tflat] = 0;
        return;
    }
    const bool prof = SC_DIAG_BIT(dbg, 4);
    const unsigned long long pc_begin = SC_DIAG_CLOCK(prof), pr_begin = SC_DIAG_REALTIME(prof);
    unsigned long long pc_stage = 0, pc_issue = 0, pc_blend = 0, pc_first = 0, pc_batches = 0, pc_splats = 0;
    int range_start, range_end;
    sc_tile_range(isect_offsets, tflat, total_tiles, n_isects, range_start, range_end);
    const int num_batches = (range_end - range_start + B - 1) / B;
    int walked = 0;                       // work done by this wave: blend iterations + 8 per staged batch (the
                                          // scheduling hint the next frame's dispatch order is built from)

    // the rectangle of pixel centres this wave owns (only pixels inside the image count)
    const float rx0 = (float)(txi * 16) + 0.5f;
    const float ry0 = (float)(tyi * 16 + sub * ROWS) + 0.5f;
    const float rx1 = (float)min(txi * 16 + 15, width - 1) + 0.5f;
    const float ry1 = (float)min(tyi * 16 + sub * ROWS + ROWS - 1, height - 1) + 0.5f;

    // Per-pixel state in PAIRS (v_pk_add / v_pk_fma / v_pk_mul process two pixels per VALU op).
    // A finished pixel (terminated, outside the image, or the unused second pixel of a quarter-tile
    // lane) is marked by poisoning its x coordinate with +inf: dx = -inf, sigma2 = +inf, alpha =
    // exp2(-inf) = 0 < 1/255, so it can never blend again and the loop needs no per-pixel `done` flag.
    // (A2 == 0 would turn that into 0 * inf = NaN, hence the staging replaces an exactly-zero A2 by
    // 1e-37, which no finite pixel can see: 1e-37 * dx^2 is absorbed by every other term.)
    const float INF = __builtin_huge_valf();
    sc_f2 pxp[NP], T2[NP];
    int cur[2 * NP];
    float acc[2 * NP][CDIM];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        pxp[p] = sc_f2{inside[2 * p] ? pxf[2 * p] : INF, inside[2 * p + 1] ? pxf[2 * p + 1] : INF};
        T2[p] = sc_f2{1.f, 1.f};
    }
#pragma unroll
    for (int k = 0; k < 2 * NP; ++k) {
        cur[k] = 0;
#pragma unroll
        for (int d = 0; d < CDIM; ++d) acc[k][d] = 0.f;
    }
    // x coordinates are positive floats or +inf, so their bit patterns order like integers
    // (integer min: no NaN canonicalisation ops)
    auto all_done = [&]() -> bool {
        int m = min(__float_as_int(pxp[0].x), __float_as_int(pxp[0].y));
        if (NP > 1) m = min(m, min(__float_as_int(pxp[NP - 1].x), __float_as_int(pxp[NP - 1].y)));
        return __all(m == 0x7f800000);
    };

    // register-staged pipeline: parameters of batch b, ids of batch b+1
    float2 p_xy[SB];
    float p_a[SB], p_b[SB], p_c[SB], p_op[SB];
    float4 p_col[SB];
    bool p_live[SB];
    int g_next[SB];
    auto load_splat = [&](int g, int j) {
        if (PACKED) {
            const float4* rec = reinterpret_cast<const float4*>(means2d) + (int64_t)g * 3;
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
            p_xy[j] = make_float2(q0.x, q0.y);
            p_a[j] = q0.z; p_b[j] = q0.w; p_c[j] = q1.x;
            p_op[j] = q1.y;
            p_col[j] = make_float4(q1.z, q1.w, q2.x, CDIM > 3 ? q2.y : 0.f);
            return;
        }
        p_xy[j] = *reinterpret_cast<const float2*>(means2d + (int64_t)g * 2);
        const float* cn = conics + (int64_t)g * 3;
        p_a[j] = cn[0]; p_b[j] = cn[1]; p_c[j] = cn[2];
        p_op[j] = opacities[g];
        const float* c = colors + (int64_t)g * CDIM;
        p_col[j] = make_float4(c[0], c[1], c[2], CDIM > 3 ? c[3] : 0.f);
    };
#pragma unroll
    for (int j = 0; j < SB; ++j) {
        const int idx0 = range_start + j * 64 + lane;
        p_live[j] = idx0 < range_end;
        p_xy[j] = make_float2(0.f, 0.f);
        p_a[j] = p_b[j] = p_c[j] = p_op[j] = 0.f;
        p_col[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p_live[j]) {
            const int g = sc_safe_id(flatten_ids[idx0], N);
            p_live[j] = g >= 0;
            if (p_live[j]) load_splat(g, j);
        }
        const int idx1 = idx0 + B;
        g_next[j] = (idx1 < range_end) ? sc_safe_id(flatten_ids[idx1], N) : -1;
    }

    for (int b = 0; b < num_batches; ++b) {
        if (all_done()) break;
        const int batch_start = range_start + B * b;
        const unsigned long long pc0 = SC_DIAG_CLOCK(prof);
        SC_DIAG_DRAIN(prof);
        const unsigned long long pcw = SC_DIAG_CLOCK(prof);
        if (prof && b == 0) pc_first = pcw - pc_begin;
        walked += 4 * SB;                 // a staged batch of 64 (gather + cull) weighs about 4 blend iterations
        // ---- cull + compact (wave-level, no workgroup barrier needed: the workgroup is this wave)
        int bsz = 0;
        __syncthreads();   // single-wave workgroup: orders the previous batch's LDS reads vs these writes
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            bool keep = false;
            if (p_live[j])
                keep = !splat_misses_rect(p_a[j], p_b[j], p_c[j], p_op[j], rx0 - p_xy[j].x, rx1 - p_xy[j].x,
                                          ry0 - p_xy[j].y, ry1 - p_xy[j].y);
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int slot = bsz + __popcll(m & sc_lanemask_lt());
                const ScSplat sp = sc_prescale(p_xy[j].x, p_xy[j].y, p_a[j], p_b[j], p_c[j], p_op[j]);
                // (mx, lop and A2 are broadcast to pixel PAIRS: they sit in even slots, the low half of a register
                // pair, which is what v_pk_* can broadcast without a move)
                xyoa_s[slot] = make_float4(sp.mx, sp.my, sp.lop, sp.B2);
                bck_s[slot] = make_float4(sp.A2 == 0.f ? 1e-37f : sp.A2, sp.C2,
                                          __int_as_float(batch_start + j * 64 + lane), 0.f);
                col_s[slot] = p_col[j];
            }
            bsz += __popcll(m);
        }
        __syncthreads();
        const unsigned long long pc1 = SC_DIAG_CLOCK(prof);
        // ---- next batch's parameters and the ids after that go in flight ---------------------------
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            p_live[j] = g_next[j] >= 0;
            if (p_live[j]) load_splat(g_next[j], j);
            const int idx2 = batch_start + 2 * B + j * 64 + lane;
            g_next[j] = (idx2 < range_end) ? sc_safe_id(flatten_ids[idx2], N) : -1;
        }
        // ---- blend ---------------------------------------------------------------------------------
        if (SC_DIAG_BIT(dbg, 1)) bsz = 0;      // diagnostic build only: price the kernel without its blend loop
        const unsigned long long pc2 = SC_DIAG_CLOCK(prof);
        if (bsz > 0) {
            // one blended splat: a = (mx, my, log2 op, B2), bc = (A2, C2, sorted index, -), c = colour
            auto blend = [&](const float4& a, const float4& bc, const float4& c) {
                const float dy = a.y - py;
                const float bdy = sc_row_b(a.w, dy), qdy = sc_row_q(bc.y, dy);    // shared by the lane's pixels
                const int sidx = __float_as_int(bc.z);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    // the pinned arithmetic of raster_common.h, two pixels per instruction
                    const sc_f2 dx = sc_f2{a.x, a.x} - pxp[p];
                    const sc_f2 tt = __builtin_elementwise_fma(sc_f2{bc.x, bc.x}, dx, sc_f2{bdy, bdy});
                    const sc_f2 sg = __builtin_elementwise_fma(tt, dx, sc_f2{qdy, qdy});
                    const sc_f2 e = sc_f2{a.z, a.z} - sg;
                    const sc_f2 al = sc_f2{fminf(SC_ALPHA_MAX, __builtin_amdgcn_exp2f(e.x)),
                                           fminf(SC_ALPHA_MAX, __builtin_amdgcn_exp2f(e.y))};
                    const bool v0 = sc_valid(sg.x, al.x), v1 = sc_valid(sg.y, al.y);
                    const sc_f2 nT = __builtin_elementwise_fma(-al, T2[p], T2[p]);
                    const bool t0 = v0 && (nT.x <= SC_T_EPS), t1 = v1 && (nT.y <= SC_T_EPS);
                    const bool b0 = v0 != t0, b1 = v1 != t1;                    // v && !t (t implies v): one mask xor
                    const sc_f2 ae = sc_f2{b0 ? al.x : 0.f, b1 ? al.y : 0.f};   // one select drives vis AND T
                    const sc_f2 vis = ae * T2[p];
                    T2[p] = __builtin_elementwise_fma(-ae, T2[p], T2[p]);       // == nT when blending, else T
                    pxp[p] = sc_f2{t0 ? INF : pxp[p].x, t1 ? INF : pxp[p].y};
                    // adding c*0 leaves the sums bit-identical to skipping (sums are never -0)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (2 * p + h >= PPL) continue;        // the unused pixel of a quarter-tile lane
                        const float vh = h ? vis.y : vis.x;
                        acc[2 * p + h][0] = __fmaf_rn(c.x, vh, acc[2 * p + h][0]);
                        acc[2 * p + h][1] = __fmaf_rn(c.y, vh, acc[2 * p + h][1]);
                        acc[2 * p + h][2] = __fmaf_rn(c.z, vh, acc[2 * p + h][2]);
                        if constexpr (CDIM > 3) acc[2 * p + h][3] = __fmaf_rn(c.w, vh, acc[2 * p + h][3]);
                    }
                    if (TRACK) {
                        cur[2 * p] = b0 ? sidx : cur[2 * p];
                        cur[2 * p + 1] = b1 ? sidx : cur[2 * p + 1];
                    }
                }
            };
            // the next record is read from LDS while the current one blends; two register sets take turns, so
            // that no record is copied from "next" to "current" (4 v_mov_b64 of ~66 VALU ops per splat)
            float4 a0 = xyoa_s[0], b0 = bck_s[0], c0 = col_s[0], a1, b1, c1;
            int t = 0;
            // (the whole-tile exit vote is taken after every SECOND splat: a splat blended onto finished pixels
            // changes nothing -- their x is +inf -- and the vote is 3 VALU + 2 SALU ops and a branch)
            for (;;) {
                a1 = xyoa_s[t + 1]; b1 = bck_s[t + 1]; c1 = col_s[t + 1];
                // The scheduler sinks these three LDS reads BELOW the blend of the current record (fewer live
                // registers), which puts their latency in front of every iteration (ISA of round 2: ds_read x3 then
                // s_waitcnt lgkmcnt(2) at the loop top).  Pinning them above costs 10 VGPRs and gives -1 us on S-1M,
                // -2..-6 us on the street scene (tools/ab_lib.py, profiles/r03_raster_prefetch_ab.txt).  With two
                // splats staged per lane the variant WITH last_ids then needed 98 VGPRs (4 waves per SIMD, +22 us) and
                // went without; with one per lane (batches of 64) every variant fits 82 and the training forward
                // gains 18 us from the pin (131 -> 113 us, profiles/r03_raster_batch_ab.txt).
                __builtin_amdgcn_sched_barrier(0);
                blend(a0, b0, c0);
                if (++t >= bsz) break;
                a0 = xyoa_s[t + 1]; b0 = bck_s[t + 1]; c0 = col_s[t + 1];
                __builtin_amdgcn_sched_barrier(0);
                blend(a1, b1, c1);
                if (all_done()) { walked += t + 1 - bsz; break; }
                if (++t >= bsz) break;
            }
            walked += bsz;
        }
        if (prof) {
            const unsigned long long pc3 = SC_DIAG_CLOCK(prof);
            pc_stage += pc2 - pc1; pc_issue += pc1 - pcw; pc_blend += pc3 - pc2; pc_batches += 1; pc_splats += (unsigned long long)bsz;
        }
    }
    if (PLANAR) {
        float Tv[PPL];
#pragma unroll
        for (int k = 0; k < PPL; ++k) Tv[k] = (k & 1) ? T2[k >> 1].y : T2[k >> 1].x;
        // (inside[] is monotone along the lane's pixels: the last one inside means all are)
        const bool v4 = PPL == 4 && (width & 3) == 0 && inside[PPL - 1];
        const bool v2 = PPL >= 2 && (width & 1) == 0;
        auto put = [&](float* base, const float* v) {
            if (PPL == 4 && v4) { *reinterpret_cast<float4*>(base) = make_float4(v[0], v[1], v[2], v[3 < PPL ? 3 : 0]); return; }
#pragma unroll
            for (int j = 0; j + 1 < PPL; j += 2) {
                if (v2 && inside[j + 1]) *reinterpret_cast<float2*>(base + j) = make_float2(v[j], v[j + 1]);
                else { if (inside[j]) base[j] = v[j]; if (inside[j + 1]) base[j + 1] = v[j + 1]; }
            }
            if (PPL == 1 && inside[0]) base[0] = v[0];
        };
        float al[PPL];
#pragma unroll
        for (int k = 0; k < PPL; ++k) al[k] = 1.0f - Tv[k];
        put(render_alphas + pix0, al);
#pragma unroll
        for (int d = 0; d < CDIM; ++d) {
            float v[PPL];
#pragma unroll
            for (int k = 0; k < PPL; ++k) v[k] = backgrounds ? acc[k][d] + Tv[k] * backgrounds[cam * CDIM + d] : acc[k][d];
            put(render_colors + ppix0 + d * plane, v);
        }
    } else
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        if (!inside[k]) continue;
        const float Tk = (k & 1) ? T2[k >> 1].y : T2[k >> 1].x;
        const int64_t pix = pix0 + k;
        render_alphas[pix] = 1.0f - Tk;
        if (CDIM == 4) {
            float4 o = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
            if (backgrounds) {
                o.x += Tk * backgrounds[cam * 4 + 0]; o.y += Tk * backgrounds[cam * 4 + 1];
                o.z += Tk * backgrounds[cam * 4 + 2]; o.w += Tk * backgrounds[cam * 4 + 3];
            }
            // "RGB+ED" epilogue (sc_rasterize_fwd_ed): expected depth = depth sum / max(alpha, 1e-10),
            // the caller's renderer.py:284 / gsplat rasterization() post-step, as one IEEE divide
            if (ED) o.w = o.w / fmaxf(1.0f - Tk, 1e-10f);
            *reinterpret_cast<float4*>(render_colors + pix * 4) = o;
        } else {
#pragma unroll
            for (int d = 0; d < CDIM; ++d)
                render_colors[pix * CDIM + d] = backgrounds ? acc[k][d] + Tk * backgrounds[cam * CDIM + d] : acc[k][d];
        }
        if (TRACK) last_ids[pix] = cur[k];
    }
    if (tile_work && lane == 0) tile_work[tflat] = walked;     // halves: the later finisher's count stands
#ifdef SC_DIAG
    const int prow = tflat * 2 + (NSUB == 1 ? 0 : sub);
    if (prof && lane == 0 && prow < SC_PHASE_ROWS) {
        const unsigned long long pc_end = SC_DIAG_CLOCK(prof);
        unsigned long long* row = g_sc_phase_cycles[prow];      // (+=: the frames of a run accumulate)
        row[0] += pc_end - pc_begin; row[1] += pc_stage; row[2] += pc_issue; row[3] += pc_blend;
        row[4] += 1ull; row[5] += pc_batches; row[6] += pc_splats; row[7] += pc_first;
        row[8] += SC_DIAG_REALTIME(prof) - pr_begin;      // the same lifetime in 100 MHz ticks: [0] / [8] = shader clock / 100 MHz
    }
#endif
}

template <int CDIM, bool TRACK, bool PACKED = false, bool ED = false, bool PLANAR = false>
__global__ __launch_bounds__(64) void raster_fwd_wave_kernel(
    const float* __restrict__ means2d, const float* __restrict__ conics,
    const float* __restrict__ colors, const float* __restrict__ opacities,
    const float* __restrict__ backgrounds, const uint8_t* __restrict__ tile_masks, int N,
    int width, int height, int tile_width, int tile_height, int total_tiles,
    const int32_t* __restrict__ isect_offsets, const int32_t* __restrict__ flatten_ids, int n_isects,
    float* __restrict__ render_colors, float* __restrict__ render_alphas,
    int32_t* __restrict__ last_ids, int map_mode, const int32_t* __restrict__ order,
    int32_t* __restrict__ tile_work SC_DIAG_PARAM(dbg)) {
    constexpr int B = 64 * RASTER_SB;
    __shared__ float4 xyoa_s[B + 1];      // mx, my, opac, conic.a      (+1: the loop prefetches t+1)
    __shared__ float4 bck_s[B + 1];       // conic.b, conic.c, sorted index (int bits), -
    __shared__ float4 col_s[B + 1];       // colour channels
    // block -> tile.  Blocks b, b + 8, b + 16, .. share an XCD (and its 4 MiB L2).
    //   map_mode 1 (default): tile = block, i.e. neighbouring tiles go round-robin over the 8 XCDs.  A dense
    //     image region (the horizon band of a street scene) is then spread over all XCDs.
    //   order (nullable): the dispatch order of the tiles, longest-running first (built by the intersection
    //     stage from tile_work, the entries every tile walked the last time: center_scatter_kernel, block 2).
    //     The launch's makespan is one tile's serial walk plus the throughput part, so the long walks start first.
    //   map_mode 0: one contiguous band of tile rows per XCD (more L2 reuse of the gathered parameters; but
    //     the XCD that owns the dense band finishes long after the others: 0.49 vs 0.33 ms on the street scene,
    //     no difference on the uniform S-1M).
    int tflat = blockIdx.x, kind = 0;
    if (order && tile_work)      // the hint bank of this call's view: the slot is the list's last word (uniform load)
        tile_work += (size_t)sc_clamp_view_slot(order[total_tiles + total_tiles / 8 + 8 + total_tiles + 1]) * total_tiles;
    if (order) {
        // an ITEM of the dispatch list (include/street_crafter_amd.h, sc_tile_order_len): tile << 2 | kind, kind 0 =
        // the whole tile, 1 / 2 = its upper / lower 16 x 8 half (two waves share a tile whose walk would otherwise be
        // the launch's tail; quarters gained nothing more: profiles/r02_raster_split_ab.txt), negative = no work
        const int item = order[blockIdx.x];
        if (item < 0) return;
        tflat = item >> 2;
        kind = item & 3;
        if (tflat >= total_tiles || kind == 3) return;
    } else {
        if (tflat >= total_tiles) return;
        if (map_mode == 0) {
            const int nwg = gridDim.x, bid = blockIdx.x;
            const int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
            tflat = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
    }
    if (kind == 0)
        raster_item<CDIM, TRACK, 1, PACKED, ED, PLANAR>(means2d, conics, colors, opacities, backgrounds, tile_masks, N, width, height,
                                    tile_width, tile_height, total_tiles, isect_offsets, flatten_ids, n_isects,
                                    render_colors, render_alphas, last_ids, tflat, 0, xyoa_s, bck_s, col_s, tile_work
                                    SC_DIAG_ARG(dbg));
    else
        raster_item<CDIM, TRACK, 2, PACKED, ED, PLANAR>(means2d, conics, colors, opacities, backgrounds, tile_masks, N, width, height,
                                    tile_width, tile_height, total_tiles, isect_offsets, flatten_ids, n_isects,
                                    render_colors, render_alphas, last_ids, tflat, kind - 1, xyoa_s, bck_s, col_s,
                                    tile_work SC_DIAG_ARG(dbg));
}

}  // namespace

int g_sc_raster_map = 1;      // sc_set_option "raster_map": block -> tile map of the wave kernel (see there)
int g_sc_raster_hint_blend = 3;   // sc_set_option "raster_hint_blend": weight (quarters) of the +-2 tile neighbourhood maximum in a tile's hint
int g_sc_raster_split = 50;   // sc_set_option "raster_split": tiles with >= this % of the heaviest tile's work are halved (0: none)

// Items of the FORWARD's dispatch list for `total_tiles` tiles: every tile once, plus room for the tiles that are split
// into halves (at most one in eight), padded with -1.
int sc_tile_order_fwd_items(int total_tiles) { return total_tiles + total_tiles / 8 + 8; }

// Length of the dispatch-list buffer: the forward's list, then the whole-tile list (same order; built only under
// raster_bwd_split 0), a word that says whether it is there, and the view slot of the call (sc_common.h).
extern "C" int sc_tile_order_len(int total_tiles) {
    return total_tiles < 0 ? 0 : sc_tile_order_fwd_items(total_tiles) + total_tiles + 2;     // + "second list present" + view slot
}

static int rasterize_fwd_impl(const float* means2d, const float* conics, const float* colors,
                              const float* opacities, const float* backgrounds,
                              const uint8_t* tile_masks, int C, int N, int D, int width, int height,
                              int tile_size, int tile_width, int tile_height,
                              const int32_t* isect_offsets, const int32_t* flatten_ids,
                              int64_t n_isects, float* render_colors, float* render_alphas,
                              int32_t* last_ids, const int32_t* tile_order, int32_t* tile_work,
                              sc_stream_t stream, int epilogue) {
    if (C < 0 || N < 0 || D < 1 || D > SC_MAX_CDIM || width <= 0 || height <= 0) return SC_EINVAL;
    if (tile_size < 1 || tile_size > 32 || tile_width <= 0 || tile_height <= 0) return SC_EINVAL;
    if (n_isects < 0 || n_isects > 0x7fffffffLL) return SC_EINVAL;
    if ((int64_t)tile_width * tile_size < width || (int64_t)tile_height * tile_size < height) return SC_EINVAL;
    if (C == 0) return SC_OK;
    if (!isect_offsets || !render_colors || !render_alphas) return SC_EINVAL;   // last_ids is nullable
    const bool packed = (epilogue & 2) != 0;      // means2d = 48-B records (raster_item<.., PACKED>)
    if (n_isects > 0 && (!means2d || !flatten_ids || (!packed && (!conics || !colors || !opacities)))) return SC_EINVAL;
    if (C > 65535 || tile_height > 65535) return SC_EINVAL;
    dim3 grid(tile_width, tile_height, C);
    if ((int64_t)C * N > 0x7fffffffLL) return SC_EINVAL;
    const int NS = C * N;                 // kernels bound-check flatten ids against C*N
    const int variant = g_sc_raster_fwd_variant;
    // the depth-normalising epilogue exists in the wave-per-tile kernel with 4 channels only
    if ((epilogue & 3) && !(variant >= 3 && tile_size == 16 && D == 4)) return SC_EUNSUPPORTED;
    if ((epilogue & 4) && !(variant >= 3 && tile_size == 16 && (D == 3 || D == 4))) return SC_EUNSUPPORTED;
    if (packed && last_ids) return SC_EUNSUPPORTED;
    const bool ed = (epilogue & 1) != 0;       // depth-normalising epilogue: a template parameter of the kernel
    const bool planar = (epilogue & 4) != 0;   // render_colors as [C][D][H][W] planes (sc_rasterize_fwd_planar)
    if (planar && (packed || ed || last_ids)) return SC_EUNSUPPORTED;
    if (variant >= 3 && tile_size == 16 && (D == 3 || D == 4)) {
        if ((int64_t)C * tile_width * tile_height >= (1 << 29)) return SC_EINVAL;
        const int total_tiles = C * tile_width * tile_height;
        const int n_blocks = tile_order ? sc_tile_order_fwd_items(total_tiles) : total_tiles;
        if (ed && last_ids) return SC_EUNSUPPORTED;       // (the epilogue is an inference form: sc_rasterize_fwd_ed passes none)
#define SC_LAUNCH_WAVE(CD, TR, PK, EDP, ...)                                                                        \
    hipLaunchKernelGGL((raster_fwd_wave_kernel<CD, TR, PK, EDP, ##__VA_ARGS__>), dim3(n_blocks), dim3(64), 0, sc_s(stream), means2d, \
                       conics, colors, opacities, backgrounds, tile_masks, NS, width, height, tile_width,           \
                       tile_height, total_tiles, isect_offsets, flatten_ids, (int)n_isects, render_colors,         \
                       render_alphas, last_ids, g_sc_raster_map, tile_order, tile_work SC_DIAG_ARG(g_sc_debug[1] & 0xff))
        if (planar) { if (D == 4) SC_LAUNCH_WAVE(4, false, false, false, true); else SC_LAUNCH_WAVE(3, false, false, false, true); }
        else if (packed) { if (ed) SC_LAUNCH_WAVE(4, false, true, true); else SC_LAUNCH_WAVE(4, false, true, false); }
        else if (D == 4) {
            if (ed) SC_LAUNCH_WAVE(4, false, false, true);
            else if (last_ids) SC_LAUNCH_WAVE(4, true, false, false);
            else SC_LAUNCH_WAVE(4, false, false, false);
        } else { if (last_ids) SC_LAUNCH_WAVE(3, true, false, false); else SC_LAUNCH_WAVE(3, false, false, false); }
#undef SC_LAUNCH_WAVE
        SC_LAUNCH_CHECK();
        return SC_OK;
    }
    dim3 block(tile_size, tile_size);
    const size_t shmem = (size_t)tile_size * tile_size * 40;
    if (D == 4)
        hipLaunchKernelGGL(raster_fwd_ref_kernel<4>, grid, block, shmem, sc_s(stream), means2d, conics, colors,
                           opacities, backgrounds, tile_masks, NS, D, width, height, tile_size, tile_width,
                           tile_height, isect_offsets, flatten_ids, (int)n_isects, render_colors,
                           render_alphas, last_ids);
    else if (D == 3)
        hipLaunchKernelGGL(raster_fwd_ref_kernel<3>, grid, block, shmem, sc_s(stream), means2d, conics, colors,
                           opacities, backgrounds, tile_masks, NS, D, width, height, tile_size, tile_width,
                           tile_height, isect_offsets, flatten_ids, (int)n_isects, render_colors,
                           render_alphas, last_ids);
    else
        hipLaunchKernelGGL(raster_fwd_ref_kernel<0>, grid, block, shmem, sc_s(stream), means2d, conics, colors,
                           opacities, backgrounds, tile_masks, NS, D, width, height, tile_size, tile_width,
                           tile_height, isect_offsets, flatten_ids, (int)n_isects, render_colors,
                           render_alphas, last_ids);
    SC_LAUNCH_CHECK();
    return SC_OK;
}

extern "C" int sc_rasterize_fwd(const float* means2d, const float* conics, const float* colors,
                                const float* opacities, const float* backgrounds,
                                const uint8_t* tile_masks, int C, int N, int D, int width, int height,
                                int tile_size, int tile_width, int tile_height,
                                const int32_t* isect_offsets, const int32_t* flatten_ids,
                                int64_t n_isects, float* render_colors, float* render_alphas,
                                int32_t* last_ids, const int32_t* tile_order, int32_t* tile_work,
                                sc_stream_t stream) {
    return rasterize_fwd_impl(means2d, conics, colors, opacities, backgrounds, tile_masks, C, N, D, width, height,
                              tile_size, tile_width, tile_height, isect_offsets, flatten_ids, n_isects,
                              render_colors, render_alphas, last_ids, tile_order, tile_work, stream, 0);
}

// render_colors stored as planes [C][D][H][W] (see raster_item<.., PLANAR>); inference form: no last_ids.  The wave kernel
// only (tile 16, 3 or 4 channels): SC_EUNSUPPORTED otherwise, and the caller takes sc_rasterize_fwd.
extern "C" int sc_rasterize_fwd_planar(const float* means2d, const float* conics, const float* colors,
                                       const float* opacities, const float* backgrounds,
                                       const uint8_t* tile_masks, int C, int N, int D, int width, int height,
                                       int tile_size, int tile_width, int tile_height,
                                       const int32_t* isect_offsets, const int32_t* flatten_ids,
                                       int64_t n_isects, float* render_colors, float* render_alphas,
                                       const int32_t* tile_order, int32_t* tile_work, sc_stream_t stream) {
    return rasterize_fwd_impl(means2d, conics, colors, opacities, backgrounds, tile_masks, C, N, D, width, height,
                              tile_size, tile_width, tile_height, isect_offsets, flatten_ids, n_isects,
                              render_colors, render_alphas, nullptr, tile_order, tile_work, stream, 4);
}

extern "C" int sc_rasterize_fwd_ed(const float* means2d, const float* conics, const float* colors,
                                   const float* opacities, const float* backgrounds,
                                   const uint8_t* tile_masks, int C, int N, int D, int width, int height,
                                   int tile_size, int tile_width, int tile_height,
                                   const int32_t* isect_offsets, const int32_t* flatten_ids,
                                   int64_t n_isects, float* render_colors, float* render_alphas,
                                   const int32_t* tile_order, int32_t* tile_work, sc_stream_t stream) {
    return rasterize_fwd_impl(means2d, conics, colors, opacities, backgrounds, tile_masks, C, N, D, width, height,
                              tile_size, tile_width, tile_height, isect_offsets, flatten_ids, n_isects,
                              render_colors, render_alphas, nullptr, tile_order, tile_work, stream, 1);
}


// The fused forward's rasterizer: `records` = one 48-B record per (camera, Gaussian) as sc_projection_sh_fwd writes
// them (x, y, conic a, b | conic c, opacity, colour 0, 1 | colour 2, 3, -, -); 4 channels, tile 16, the wave kernel.
// depth_normalise != 0: the 4th channel is divided by max(alpha, 1e-10) as in sc_rasterize_fwd_ed.
extern "C" int sc_rasterize_fwd_packed(const float* records, const float* backgrounds, const uint8_t* tile_masks,
                                       int C, int N, int width, int height, int tile_width, int tile_height,
                                       const int32_t* isect_offsets, const int32_t* flatten_ids, int64_t n_isects,
                                       float* render_colors, float* render_alphas, const int32_t* tile_order,
                                       int32_t* tile_work, int depth_normalise, sc_stream_t stream) {
    if (g_sc_raster_fwd_variant < 3) return SC_EUNSUPPORTED;
    if (records && ((uintptr_t)records & 15)) return SC_EINVAL;
    return rasterize_fwd_impl(records, nullptr, nullptr, nullptr, backgrounds, tile_masks, C, N, 4, width, height, 16,
                              tile_width, tile_height, isect_offsets, flatten_ids, n_isects, render_colors,
                              render_alphas, nullptr, tile_order, tile_work, stream,
                              2 | (depth_normalise ? 1 : 0));
}
