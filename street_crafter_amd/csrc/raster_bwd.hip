// a11: rasterize_to_pixels backward for gfx950.
// Reached from the reference through loss.backward() (train.py:236); the absgrad output is what
// street_gaussian/models/street_gaussian_model.py:505-506 reads as `means2d.absgrad`.
//
// One workgroup per tile, one lane per pixel; the tile's splat list is replayed back to front
// from the last splat any pixel of the wave blended.  Per splat the 64 lanes' contributions are
// summed across the wave first, then one lane issues the global float atomics (gradient
// buffers must arrive zero-filled).  Float atomics make the result order-dependent in the last
// bits; tests compare against the autograd oracle with a tolerance.
#include "raster_common.h"

int g_sc_raster_bwd_variant = 1;   // 0 = reference-shaped, 1 = one wave per tile (default)
int g_sc_raster_bwd_split = 1;     // 1 = the backward follows the forward's dispatch list incl. its half tiles, 0 = whole tiles only

namespace {

template <int CDIM>
__global__ void raster_bwd_kernel(
    const float* __restrict__ means2d, const float* __restrict__ conics,
    const float* __restrict__ colors, const float* __restrict__ opacities,
    const float* __restrict__ backgrounds, const uint8_t* __restrict__ tile_masks, int N, int D,
    int width, int height, int tile_size, int tile_width, int tile_height,
    const int32_t* __restrict__ isect_offsets, const int32_t* __restrict__ flatten_ids, int n_isects,
    const float* __restrict__ render_alphas, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_render_colors, const float* __restrict__ v_render_alphas,
    float* __restrict__ v_means2d_abs, float* __restrict__ v_means2d, float* __restrict__ v_conics,
    float* __restrict__ v_colors, float* __restrict__ v_opacities) {
    constexpr int ND = CDIM > 0 ? CDIM : SC_MAX_CDIM;
    extern __shared__ __align__(16) unsigned char smem[];
    const int B = blockDim.x * blockDim.y;
    int* id_s = reinterpret_cast<int*>(smem);                              // [B]
    float4* xyoa_s = reinterpret_cast<float4*>(smem + (size_t)B * 16);     // [B]
    float2* bc_s = reinterpret_cast<float2*>(smem + (size_t)B * 32);       // [B]
    float* rgb_s = reinterpret_cast<float*>(smem + (size_t)B * 40);        // [B * D]

    const int cam = blockIdx.z;
    const int tile_id = blockIdx.y * tile_width + blockIdx.x;
    const int tflat = cam * tile_width * tile_height + tile_id;
    if (tile_masks && !tile_masks[tflat]) return;
    const int px_i = blockIdx.x * tile_size + threadIdx.x;
    const int py_i = blockIdx.y * tile_size + threadIdx.y;
    const float px = (float)px_i + 0.5f, py = (float)py_i + 0.5f;
    const bool inside = (px_i < width) && (py_i < height);
    const int64_t pix = inside ? ((int64_t)cam * height + py_i) * width + px_i : 0;
    const int tr = threadIdx.y * blockDim.x + threadIdx.x;
    const int lane = tr & 63;

    const int total_tiles = gridDim.z * tile_width * tile_height;
    int range_start, range_end;
    sc_tile_range(isect_offsets, tflat, total_tiles, n_isects, range_start, range_end);
    const int num_batches = (range_end - range_start + B - 1) / B;

    const float T_final = inside ? 1.0f - render_alphas[pix] : 1.0f;
    float T = T_final;
    float buffer[ND];
    float v_rc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        buffer[d] = 0.f;
        v_rc[d] = (inside && (CDIM > 0 || d < D)) ? v_render_colors[pix * D + d] : 0.f;
    }
    const float v_ra = inside ? v_render_alphas[pix] : 0.f;
    float bg_dot = 0.f;
    if (backgrounds) {
#pragma unroll
        for (int d = 0; d < ND; ++d)
            if (CDIM > 0 || d < D) bg_dot += backgrounds[cam * D + d] * v_rc[d];
    }
    const int bin_final = inside ? last_ids[pix] : 0;
    int wave_bin_final = bin_final;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wave_bin_final = max(wave_bin_final, __shfl_xor(wave_bin_final, o, 64));

    for (int b = 0; b < num_batches; ++b) {
        __syncthreads();
        const int batch_end = range_end - 1 - B * b;
        const int bsz = min(B, batch_end + 1 - range_start);
        const int idx = batch_end - tr;
        if (idx >= range_start) {
            const int g = sc_safe_id(flatten_ids[idx], N);
            if (g >= 0) {
                id_s[tr] = g;
                const float2 xy = *reinterpret_cast<const float2*>(means2d + (int64_t)g * 2);
                const float* cn = conics + (int64_t)g * 3;
                xyoa_s[tr] = make_float4(xy.x, xy.y, opacities[g], cn[0]);
                bc_s[tr] = make_float2(cn[1], cn[2]);
                for (int d = 0; d < D; ++d) rgb_s[tr * D + d] = colors[(int64_t)g * D + d];
            } else {                                   // dead entry: opacity 0 -> alpha 0 -> skipped
                id_s[tr] = 0;
                xyoa_s[tr] = make_float4(0.f, 0.f, 0.f, 0.f);
                bc_s[tr] = make_float2(0.f, 0.f);
                for (int d = 0; d < D; ++d) rgb_s[tr * D + d] = 0.f;
            }
        }
        __syncthreads();
        for (int t = max(0, batch_end - wave_bin_final); t < bsz; ++t) {
            bool valid = inside && (batch_end - t <= bin_final);
            float alpha = 0.f, opac = 0.f, vis = 0.f, dx = 0.f, dy = 0.f;
            float ca = 0.f, cb = 0.f, cc = 0.f;
            if (valid) {
                const float4 a = xyoa_s[t];
                const float2 bc = bc_s[t];
                ca = a.w; cb = bc.x; cc = bc.y; opac = a.z;
                dx = a.x - px; dy = a.y - py;
                // the skip decision uses the forward's exact arithmetic (raster_common.h) ...
                const ScSplat sp = sc_prescale(a.x, a.y, ca, cb, cc, opac);
                const float sigma2 = sc_sigma2(sp.A2, sc_row_b(sp.B2, dy), sc_row_q(sp.C2, dy), dx);
                alpha = sc_alpha2(sp.lop, sigma2);
                if (!sc_valid(sigma2, alpha)) valid = false;
                // ... the gradient formulas need exp(-sigma) itself
                vis = __builtin_amdgcn_exp2f(-sigma2);
            }
            if (!__any(valid)) continue;
            float v_rgb[ND];
#pragma unroll
            for (int d = 0; d < ND; ++d) v_rgb[d] = 0.f;
            float v_con0 = 0.f, v_con1 = 0.f, v_con2 = 0.f, v_x = 0.f, v_y = 0.f, v_xa = 0.f, v_ya = 0.f, v_op = 0.f;
            if (valid) {
                const float ra = 1.0f / (1.0f - alpha);
                T *= ra;
                const float fac = alpha * T;
                float v_alpha = 0.f;
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    if (CDIM > 0 || d < D) {
                        const float c = rgb_s[t * D + d];
                        v_rgb[d] = fac * v_rc[d];
                        v_alpha += (c * T - buffer[d] * ra) * v_rc[d];
                        buffer[d] += c * fac;
                    }
                }
                v_alpha += T_final * ra * v_ra;
                if (backgrounds) v_alpha += -T_final * ra * bg_dot;
                if (opac * vis <= SC_ALPHA_MAX) {
                    const float v_sigma = -opac * vis * v_alpha;
                    v_con0 = 0.5f * v_sigma * dx * dx;
                    v_con1 = v_sigma * dx * dy;
                    v_con2 = 0.5f * v_sigma * dy * dy;
                    v_x = v_sigma * (ca * dx + cb * dy);
                    v_y = v_sigma * (cb * dx + cc * dy);
                    v_xa = fabsf(v_x);
                    v_ya = fabsf(v_y);
                    v_op = vis * v_alpha;
                }
            }
#pragma unroll
            for (int d = 0; d < ND; ++d)
                if (CDIM > 0 || d < D) v_rgb[d] = sc_wave_sum(v_rgb[d]);
            v_con0 = sc_wave_sum(v_con0); v_con1 = sc_wave_sum(v_con1); v_con2 = sc_wave_sum(v_con2);
            v_x = sc_wave_sum(v_x); v_y = sc_wave_sum(v_y); v_op = sc_wave_sum(v_op);
            if (v_means2d_abs) { v_xa = sc_wave_sum(v_xa); v_ya = sc_wave_sum(v_ya); }
            if (lane == 0) {
                const int64_t g = id_s[t];
#pragma unroll
                for (int d = 0; d < ND; ++d)
                    if (CDIM > 0 || d < D) atomicAdd(v_colors + g * D + d, v_rgb[d]);
                atomicAdd(v_conics + g * 3 + 0, v_con0);
                atomicAdd(v_conics + g * 3 + 1, v_con1);
                atomicAdd(v_conics + g * 3 + 2, v_con2);
                atomicAdd(v_means2d + g * 2 + 0, v_x);
                atomicAdd(v_means2d + g * 2 + 1, v_y);
                if (v_means2d_abs) {
                    atomicAdd(v_means2d_abs + g * 2 + 0, v_xa);
                    atomicAdd(v_means2d_abs + g * 2 + 1, v_ya);
                }
                atomicAdd(v_opacities + g, v_op);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Wave-per-tile backward (tile 16x16, CDIM 3/4): the mirror image of raster_fwd_wave_kernel.
//   * one wave per tile, 4 pixels per lane: each lane first sums the contributions of its own 4
//     pixels, so the cross-lane reduction per splat happens once per TILE (the reference-shaped
//     kernel above reduces once per 64 pixels, i.e. four times per tile);
//   * the (up to) 12 per-splat sums are reduced TOGETHER by a transposing butterfly
//     (wave_transpose_sum16): every stage halves the number of live registers while it halves the
//     lanes that own a given sum - v_permlane32_swap / v_permlane16_swap (gfx950) for the two
//     cross-row stages, DPP row_mirror / row_half_mirror / quad_perm for the rest.  ~35 VALU ops
//     for all sums instead of 6 per sum, no LDS round trips, and lane 4 i ends up owning sum i, so
//     ONE global_atomic_add_f32 with 12 active lanes replaces 12 single-lane atomics;
//   * batches are staged back to front from the last splat any pixel of the tile blended, with the
//     forward's exact tile-level cull + ballot compaction, so splats that touch no pixel of the tile
//     (66 % of the walked ones on S-1M) cost one lane-test instead of 256 pixel evaluations;
//   * skip / blend decisions use the forward's pinned arithmetic (raster_common.h).
// ------------------------------------------------------------------------------------------
// lanes 0..31 get a(l) + a(l + 32), lanes 32..63 get b(l - 32) + b(l)
__device__ __forceinline__ float swap32_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// even rows get a(l) + a(l + 16), odd rows get b(l - 16) + b(l)   (row = 16 lanes)
__device__ __forceinline__ float swap16_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// 16 per-lane partial sums in, one register out: lane l holds the 64-lane total of v[l >> 2].
// Lane pairings per stage: l^32, l^16, l^15 (row_mirror), l^7 (row_half_mirror), l^2, l^1 - six
// independent masks, so every lane's contribution reaches the owner of each sum exactly once.
__device__ __forceinline__ float wave_transpose_sum16(const float (&v)[16], int lane) {
    float w[8], x[4], y[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = swap32_add(v[i], v[i + 8]);       // lane bit 5 picks +8
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = swap16_add(w[i], w[i + 4]);       // lane bit 4 picks +4
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                         // lane bit 3 picks +2
        const float keep = b3 ? x[i + 2] : x[i], give = b3 ? x[i] : x[i + 2];
        y[i] = keep + dpp_move<0x140>(give);                              // row_mirror
    }
    const float keep = b2 ? y[1] : y[0], give = b2 ? y[0] : y[1];       // lane bit 2 picks +1
    float z = keep + dpp_move<0x141>(give);                               // row_half_mirror
    z += dpp_move<0x4E>(z);                                               // quad_perm [2,3,0,1]
    z += dpp_move<0xB1>(z);                                               // quad_perm [1,0,3,2]
    return z;
}

// One wave's share of a tile: the whole 16 x 16 tile (NSUB 1, 4 pixels per lane) or its upper / lower 16 x 8 half
// (NSUB 2, `sub` 0 / 1, 2 pixels per lane: the forward's dispatch list splits the tiles whose walk would be the
// launch's tail, and the backward's walk is ~2.5x the forward's).  Both halves add into the same gradients.
template <int CDIM, int NSUB>
__device__ __forceinline__ void raster_bwd_item(
    const float* __restrict__ means2d, const float* __restrict__ conics,
    const float* __restrict__ colors, const float* __restrict__ opacities,
    const float* __restrict__ backgrounds, const uint8_t* __restrict__ tile_masks, int N,
    int width, int height, int tile_width, int tile_height, int total_tiles,
    const int32_t* __restrict__ isect_offsets, const int32_t* __restrict__ flatten_ids, int n_isects,
    const float* __restrict__ render_alphas, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_render_colors, const float* __restrict__ v_render_alphas,
    float* __restrict__ v_means2d_abs, float* __restrict__ v_means2d, float* __restrict__ v_conics,
    float* __restrict__ v_colors, float* __restrict__ v_opacities, int tflat, int sub,
    float4* xyoa_s, float4* bck_s, float4* col_s) {
    constexpr int SB = 2;      // (one per lane, which pays in the forward, changes nothing here: 285.6 vs 287.6 us)
    constexpr int B = 64 * SB;
    if (tile_masks && !tile_masks[tflat]) return;
    const int tiles_per_cam = tile_width * tile_height;
    const int cam = tflat / tiles_per_cam;
    const int tile_id = tflat - cam * tiles_per_cam;
    const int tyi = tile_id / tile_width, txi = tile_id - tyi * tile_width;
    const int lane = threadIdx.x;
    constexpr int NP = NSUB == 1 ? 2 : 1;          // pixel PAIRS per lane (4 or 2 pixels)
    constexpr int PPL = 2 * NP, LPR = 16 / PPL, ROWS = 16 / NSUB;     // lanes per row, rows this wave covers
    const int px0_i = txi * 16 + PPL * (lane % LPR), py_i = tyi * 16 + sub * ROWS + lane / LPR;
    const float py = (float)py_i + 0.5f;
    int range_start, range_end;
    sc_tile_range(isect_offsets, tflat, total_tiles, n_isects, range_start, range_end);
    if (range_end <= range_start) return;

    // per-pixel state in pairs (pixels 2p, 2p+1): x centre, running transmittance, W (below), upstream colour gradients.
    // A.6 keeps the colour sums BEHIND the current splat per channel (buf_d) and forms
    //     v_alpha = sum_d (c_d T - buf_d / (1 - alpha)) v_d + T_final (v_a - bg.v) / (1 - alpha)
    // per splat: five multiply-adds per channel and pixel.  Only the dot product P = sum_d buf_d v_d is ever used, and
    // buf_d += c_d fac means P += fac Q with Q = sum_d c_d v_d, so ONE running value per pixel does:
    //     W = T_final (v_a - bg.v) - P,      v_alpha = W / (1 - alpha) + T Q,      W -= fac Q
    // (round 3: 21 -> 11 packed operations per pixel pair and splat, 12 fewer VGPRs; the sums are the same terms in
    // another order, so gradients move at rounding level only: tests compare with the float64 autograd oracle and with
    // the reference-shaped kernel, which keeps A.6's literal form)
    sc_f2 pxp[NP], T2[NP], W2[NP], vrc[NP][CDIM];
    int bin_final[PPL];
    bool ins[PPL];
    int tile_last = -1;
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int p = k >> 1, h = k & 1;
        ins[k] = (px0_i + k < width) && (py_i < height);
        const int64_t pix = ((int64_t)cam * height + py_i) * width + px0_i + k;
        const float T_fin = ins[k] ? 1.0f - render_alphas[pix] : 1.0f;
        const float v_ra = ins[k] ? v_render_alphas[pix] : 0.f;
        bin_final[k] = ins[k] ? last_ids[pix] : -1;
        float bgdot = 0.f;
        pxp[p][h] = (float)(px0_i + k) + 0.5f;
        T2[p][h] = T_fin;
#pragma unroll
        for (int d = 0; d < CDIM; ++d) {
            const float g = ins[k] ? v_render_colors[pix * CDIM + d] : 0.f;
            vrc[p][d][h] = g;
            if (backgrounds) bgdot += backgrounds[cam * CDIM + d] * g;
        }
        W2[p][h] = T_fin * (v_ra - bgdot);
        tile_last = max(tile_last, bin_final[k]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) tile_last = max(tile_last, __shfl_xor(tile_last, o, 64));
    tile_last = min(tile_last, range_end - 1);
    if (tile_last < range_start) return;

    const float rx0 = (float)(txi * 16) + 0.5f;
    const float ry0 = (float)(tyi * 16 + sub * ROWS) + 0.5f;
    const float rx1 = (float)min(txi * 16 + 15, width - 1) + 0.5f;
    const float ry1 = (float)min(tyi * 16 + sub * ROWS + ROWS - 1, height - 1) + 0.5f;
    constexpr float LN2 = 0.6931471805599453f;

    // lane 4 i owns reduced sum i (wave_transpose_sum16): 0..3 colour channels, 4..6 conic, 7..8 mean,
    // 9..10 |mean| (absgrad), 11 opacity.  out_base == nullptr: this lane issues no atomic.
    float* out_base = nullptr;
    int out_stride = 0;
    {
        const int vi = lane >> 2;
        if ((lane & 3) == 0) {
            if (vi < CDIM) { out_base = v_colors + vi; out_stride = CDIM; }
            else if (vi >= 4 && vi <= 6) { out_base = v_conics + (vi - 4); out_stride = 3; }
            else if (vi == 7 || vi == 8) { out_base = v_means2d + (vi - 7); out_stride = 2; }
            else if ((vi == 9 || vi == 10) && v_means2d_abs) { out_base = v_means2d_abs + (vi - 9); out_stride = 2; }
            else if (vi == 11) { out_base = v_opacities; out_stride = 1; }
        }
    }

    for (int hi = tile_last; hi >= range_start; hi -= B) {
        // ---- stage (descending sorted index), cull, compact ----------------------------------------
        // Same shape as raster_fwd_wave_kernel's staging: parameters land in plain scalars, the cull
        // result is one bool, and the record is built at the LDS store.  (Carrying whole float4
        // records through the `if (keep)` merge was miscompiled by hipcc 7.2 for gfx950: the j = 1
        // record lost its mean; tests/test_gpu_parity.py::test_rasterize_backward_wave_matches_reference
        // guards against a recurrence.)
        int bsz = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            const int idx = hi - (j * 64 + lane);
            const int g = (idx >= range_start) ? sc_safe_id(flatten_ids[idx], N) : -1;
            float2 xy = make_float2(0.f, 0.f);
            float ca = 0.f, cb = 0.f, cc = 0.f, op = 0.f;
            bool keep = false;
            if (g >= 0) {
                xy = *reinterpret_cast<const float2*>(means2d + (int64_t)g * 2);
                const float* cn = conics + (int64_t)g * 3;
                ca = cn[0]; cb = cn[1]; cc = cn[2];
                op = opacities[g];
                const bool miss = splat_misses_rect(ca, cb, cc, op, rx0 - xy.x, rx1 - xy.x, ry0 - xy.y, ry1 - xy.y);
                keep = !miss;
            }
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int slot = bsz + __popcll(m & sc_lanemask_lt());
                const ScSplat sp = sc_prescale(xy.x, xy.y, ca, cb, cc, op);
                const float* c = colors + (int64_t)g * CDIM;
                xyoa_s[slot] = make_float4(sp.mx, sp.my, sp.lop, sp.A2);
                bck_s[slot] = make_float4(sp.B2, sp.C2, __int_as_float(idx), __int_as_float(g));
                col_s[slot] = make_float4(c[0], c[1], c[2], CDIM > 3 ? c[3] : 0.f);
            }
            bsz += __popcll(m);
        }
        __syncthreads();
        // ---- replay -----------------------------------------------------------------------------------
        // (the next record is read from LDS while the current one is replayed; two register sets take turns)
        auto replay = [&](const float4& a, const float4& bc, const float4& c) {
            const float cl[4] = {c.x, c.y, c.z, c.w};
            const int sidx = __float_as_int(bc.z);
            const float dy = a.y - py;
            const float bdy = sc_row_b(bc.x, dy), qdy = sc_row_q(bc.y, dy);
            // Two pixels per VALU op (v_pk_*).  Per-pixel work is kept to what cannot be factored out:
            // with v_sigma = dL/dsigma2-ish of a pixel and dx its offset, the conic, mean and opacity
            // gradients of the splat only need  Sv = sum v_sigma,  Sx = sum v_sigma dx,  Sxx = sum
            // v_sigma dx^2  (dy, A2, B2, C2, 1/op are the same for the lane's four pixels), so they are
            // assembled once per splat below; only |grad mean| (absgrad) needs the per-pixel values.
            sc_f2 Sc[CDIM], Sv = {0.f, 0.f}, Sx = {0.f, 0.f}, Sxx = {0.f, 0.f};
#pragma unroll
            for (int d = 0; d < CDIM; ++d) Sc[d] = sc_f2{0.f, 0.f};
            float s_xa = 0.f, s_ya = 0.f;
            bool any_valid = false;
            const float bdy_ln2 = LN2 * bc.x * dy, a2_ln2 = LN2 * 2.0f * a.w;       // d sigma / d mean_x pieces
            const float b2_ln2 = LN2 * bc.x, c2dy_ln2 = LN2 * 2.0f * bc.y * dy;     // d sigma / d mean_y pieces
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const sc_f2 dx = sc_f2{a.x, a.x} - pxp[p];
                const sc_f2 sg = __builtin_elementwise_fma(
                    __builtin_elementwise_fma(sc_f2{a.w, a.w}, dx, sc_f2{bdy, bdy}), dx, sc_f2{qdy, qdy});
                const sc_f2 e = sc_f2{a.z, a.z} - sg;
                const sc_f2 araw = sc_f2{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};   // op exp(-sigma)
                const sc_f2 al = sc_f2{fminf(SC_ALPHA_MAX, araw.x), fminf(SC_ALPHA_MAX, araw.y)};
                // (a pixel outside the image has bin_final = -1 < every sorted index: no separate `inside` test)
                const bool v0 = (sidx <= bin_final[2 * p]) && sc_valid(sg.x, al.x);
                const bool v1 = (sidx <= bin_final[2 * p + 1]) && sc_valid(sg.y, al.y);
                any_valid = any_valid || v0 || v1;
                const sc_f2 om = sc_f2{1.0f, 1.0f} - al;                              // >= 1e-3
                const sc_f2 ra = sc_f2{__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
                const sc_f2 Tn = T2[p] * ra;                                          // transmittance in front
                const sc_f2 at = al * Tn;
                const sc_f2 fac = sc_f2{v0 ? at.x : 0.f, v1 ? at.y : 0.f};
                sc_f2 Q = sc_f2{cl[0], cl[0]} * vrc[p][0];                            // sum_d c_d v_d of this splat
#pragma unroll
                for (int d = 1; d < CDIM; ++d) Q = __builtin_elementwise_fma(sc_f2{cl[d], cl[d]}, vrc[p][d], Q);
#pragma unroll
                for (int d = 0; d < CDIM; ++d) Sc[d] = __builtin_elementwise_fma(fac, vrc[p][d], Sc[d]);
                const sc_f2 va = __builtin_elementwise_fma(ra, W2[p], Tn * Q);        // W / (1 - alpha) + T Q
                W2[p] = __builtin_elementwise_fma(-fac, Q, W2[p]);                    // the splat moves behind: P += fac Q
                // sigma = sigma2 ln2;  d alpha / d sigma = -alpha_raw (only while alpha is not clamped)
                const bool l0 = v0 && araw.x <= SC_ALPHA_MAX, l1 = v1 && araw.y <= SC_ALPHA_MAX;
                const sc_f2 vsr = araw * va;
                const sc_f2 vs = sc_f2{l0 ? -vsr.x : 0.f, l1 ? -vsr.y : 0.f};
                const sc_f2 vd = vs * dx;
                Sv += vs;
                Sx += vd;
                Sxx = __builtin_elementwise_fma(vd, dx, Sxx);
                if (v_means2d_abs) {                                                  // uniform
                    // |d L / d mean| of a pixel = |v_sigma| |d sigma / d mean|: the second factor needs no v_sigma,
                    // and the absolute values are source modifiers of the accumulating fma (no separate abs / add)
                    const sc_f2 gxf = __builtin_elementwise_fma(sc_f2{a2_ln2, a2_ln2}, dx, sc_f2{bdy_ln2, bdy_ln2});
                    const sc_f2 gyf = __builtin_elementwise_fma(sc_f2{b2_ln2, b2_ln2}, dx, sc_f2{c2dy_ln2, c2dy_ln2});
                    s_xa = __fmaf_rn(fabsf(vs.x), fabsf(gxf.x), s_xa);
                    s_xa = __fmaf_rn(fabsf(vs.y), fabsf(gxf.y), s_xa);
                    s_ya = __fmaf_rn(fabsf(vs.x), fabsf(gyf.x), s_ya);
                    s_ya = __fmaf_rn(fabsf(vs.y), fabsf(gyf.y), s_ya);
                }
                T2[p] = sc_f2{v0 ? Tn.x : T2[p].x, v1 ? Tn.y : T2[p].y};
            }
            if (!__any(any_valid)) return;
            const float sv = Sv.x + Sv.y, sx = Sx.x + Sx.y, sxx = Sxx.x + Sxx.y;
            float s[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
            for (int d = 0; d < CDIM; ++d) s[d] = Sc[d].x + Sc[d].y;
            s[4] = 0.5f * sxx;                       // d sigma / d conic = (dx^2 / 2, dx dy, dy^2 / 2)
            s[5] = dy * sx;
            s[6] = 0.5f * dy * dy * sv;
            s[7] = a2_ln2 * sx + bdy_ln2 * sv;       // d sigma / d mean = ln2 (2 A2 dx + B2 dy, B2 dx + 2 C2 dy)
            s[8] = b2_ln2 * sx + c2dy_ln2 * sv;
            s[9] = s_xa; s[10] = s_ya;
            s[11] = -__builtin_amdgcn_exp2f(-a.z) * sv;   // d alpha / d op = alpha_raw / op, i.e. -v_sigma / op
            const float total = wave_transpose_sum16(s, lane);
            if (out_base) atomicAdd(out_base + (int64_t)__float_as_int(bc.w) * out_stride, total);
        };
        if (bsz > 0) {
            float4 a0 = xyoa_s[0], b0 = bck_s[0], c0 = col_s[0], a1, b1, c1;
            for (int t = 0;;) {
                a1 = xyoa_s[t + 1]; b1 = bck_s[t + 1]; c1 = col_s[t + 1];
                replay(a0, b0, c0);
                if (++t >= bsz) break;
                a0 = xyoa_s[t + 1]; b0 = bck_s[t + 1]; c0 = col_s[t + 1];
                replay(a1, b1, c1);
                if (++t >= bsz) break;
            }
        }
    }
}

template <int CDIM>
__global__ __launch_bounds__(64) void raster_bwd_wave_kernel(
    const float* __restrict__ means2d, const float* __restrict__ conics,
    const float* __restrict__ colors, const float* __restrict__ opacities,
    const float* __restrict__ backgrounds, const uint8_t* __restrict__ tile_masks, int N,
    int width, int height, int tile_width, int tile_height, int total_tiles,
    const int32_t* __restrict__ isect_offsets, const int32_t* __restrict__ flatten_ids, int n_isects,
    const float* __restrict__ render_alphas, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_render_colors, const float* __restrict__ v_render_alphas,
    float* __restrict__ v_means2d_abs, float* __restrict__ v_means2d, float* __restrict__ v_conics,
    float* __restrict__ v_colors, float* __restrict__ v_opacities, int map_mode,
    const int32_t* __restrict__ order, const int32_t* __restrict__ list_ok) {
    constexpr int B = 128;
    __shared__ float4 xyoa_s[B + 1];      // mx, my, log2(op), A2
    __shared__ float4 bck_s[B + 1];       // B2, C2, sorted index (int bits), flat id (int bits)
    __shared__ float4 col_s[B + 1];

    int tflat = blockIdx.x, kind = 0;     // block -> tile map: as raster_fwd_wave_kernel ("raster_map", order)
    if (order) {
        // an item of a dispatch list (include/street_crafter_amd.h): tile << 2 | kind, kind 0 = the whole tile,
        // 1 / 2 = its upper / lower half, negative = no work.  Which of the buffer's two lists the kernel is
        // given (the forward's, with halves, or the whole-tile one behind it) is the host's choice.
        // (list_ok: for the whole-tile list, the word behind it says whether the intersection stage built it --
        // it does under raster_bwd_split 0 only; without it the tiles are taken in their own order)
        const int item = (list_ok == nullptr || *list_ok != 0) ? order[blockIdx.x] : (int)blockIdx.x << 2;
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
        raster_bwd_item<CDIM, 1>(means2d, conics, colors, opacities, backgrounds, tile_masks, N, width, height, tile_width,
                                 tile_height, total_tiles, isect_offsets, flatten_ids, n_isects, render_alphas, last_ids,
                                 v_render_colors, v_render_alphas, v_means2d_abs, v_means2d, v_conics, v_colors,
                                 v_opacities, tflat, 0, xyoa_s, bck_s, col_s);
    else
        raster_bwd_item<CDIM, 2>(means2d, conics, colors, opacities, backgrounds, tile_masks, N, width, height, tile_width,
                                 tile_height, total_tiles, isect_offsets, flatten_ids, n_isects, render_alphas, last_ids,
                                 v_render_colors, v_render_alphas, v_means2d_abs, v_means2d, v_conics, v_colors,
                                 v_opacities, tflat, kind - 1, xyoa_s, bck_s, col_s);
}

// unit-test hook for wave_transpose_sum16: in [n_waves][16][64], out [n_waves][64]
__global__ __launch_bounds__(64) void test_wave_transpose_kernel(const float* __restrict__ in, float* __restrict__ out) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = in[((int64_t)blockIdx.x * 16 + i) * 64 + threadIdx.x];
    out[(int64_t)blockIdx.x * 64 + threadIdx.x] = wave_transpose_sum16(v, threadIdx.x);
}

}  // namespace

extern "C" int sc_test_wave_transpose_sum16(const float* in, int n_waves, float* out, sc_stream_t stream) {
    if (n_waves < 0) return SC_EINVAL;
    if (n_waves == 0) return SC_OK;
    if (!in || !out) return SC_EINVAL;
    hipLaunchKernelGGL(test_wave_transpose_kernel, dim3(n_waves), dim3(64), 0, sc_s(stream), in, out);
    SC_LAUNCH_CHECK();
    return SC_OK;
}

extern "C" int sc_rasterize_bwd(const float* means2d, const float* conics, const float* colors,
                                const float* opacities, const float* backgrounds,
                                const uint8_t* tile_masks, int C, int N, int D, int width, int height,
                                int tile_size, int tile_width, int tile_height,
                                const int32_t* isect_offsets, const int32_t* flatten_ids,
                                int64_t n_isects, const float* render_alphas, const int32_t* last_ids,
                                const float* v_render_colors, const float* v_render_alphas,
                                float* v_means2d_abs, float* v_means2d, float* v_conics,
                                float* v_colors, float* v_opacities, const int32_t* tile_order, sc_stream_t stream) {
    if (C < 0 || N < 0 || D < 1 || D > SC_MAX_CDIM || width <= 0 || height <= 0) return SC_EINVAL;
    if (tile_size < 1 || tile_size > 32 || tile_width <= 0 || tile_height <= 0) return SC_EINVAL;
    if (n_isects < 0 || n_isects > 0x7fffffffLL) return SC_EINVAL;
    if ((int64_t)tile_width * tile_size < width || (int64_t)tile_height * tile_size < height) return SC_EINVAL;
    if (C == 0 || n_isects == 0) return SC_OK;
    if (!means2d || !conics || !colors || !opacities || !isect_offsets || !flatten_ids || !render_alphas ||
        !last_ids || !v_render_colors || !v_render_alphas || !v_means2d || !v_conics || !v_colors ||
        !v_opacities)
        return SC_EINVAL;
    if (C > 65535 || tile_height > 65535) return SC_EINVAL;
    if ((int64_t)C * N > 0x7fffffffLL) return SC_EINVAL;
    if (g_sc_raster_bwd_variant == 1 && tile_size == 16 && (D == 3 || D == 4)) {
        if ((int64_t)C * tile_width * tile_height >= (1 << 29)) return SC_EINVAL;
        const int total_tiles = C * tile_width * tile_height;
        // with a dispatch-list buffer: its forward list (halves included: the backward's long walks are cut like the
        // forward's) or the whole-tile list behind it (sc_set_option "raster_bwd_split" 0)
        const bool halves = tile_order && g_sc_raster_bwd_split;
        const int n_blocks = halves ? sc_tile_order_fwd_items(total_tiles) : total_tiles;
        const int32_t* bwd_order = !tile_order ? nullptr : (halves ? tile_order : tile_order + sc_tile_order_fwd_items(total_tiles));
        const int32_t* list_ok = (tile_order && !halves) ? tile_order + sc_tile_order_fwd_items(total_tiles) + total_tiles : nullptr;
#define SC_LAUNCH_BWD_WAVE(CD)                                                                                  \
    hipLaunchKernelGGL(raster_bwd_wave_kernel<CD>, dim3(n_blocks), dim3(64), 0, sc_s(stream), means2d, conics,      \
                       colors, opacities, backgrounds, tile_masks, C * N, width, height, tile_width, tile_height,  \
                       total_tiles, isect_offsets, flatten_ids, (int)n_isects, render_alphas, last_ids,            \
                       v_render_colors, v_render_alphas, v_means2d_abs, v_means2d, v_conics, v_colors, v_opacities,    \
                       g_sc_raster_map, bwd_order, list_ok)
        if (D == 4) SC_LAUNCH_BWD_WAVE(4);
        else SC_LAUNCH_BWD_WAVE(3);
#undef SC_LAUNCH_BWD_WAVE
        SC_LAUNCH_CHECK();
        return SC_OK;
    }
    dim3 grid(tile_width, tile_height, C), block(tile_size, tile_size);
    const size_t B = (size_t)tile_size * tile_size;
    const size_t shmem = B * 40 + B * D * 4;
#define SC_LAUNCH_BWD(CD)                                                                              \
    hipLaunchKernelGGL(raster_bwd_kernel<CD>, grid, block, shmem, sc_s(stream), means2d, conics, colors,   \
                       opacities, backgrounds, tile_masks, C * N, D, width, height, tile_size, tile_width, \
                       tile_height, isect_offsets, flatten_ids, (int)n_isects, render_alphas, last_ids,    \
                       v_render_colors, v_render_alphas, v_means2d_abs, v_means2d, v_conics, v_colors,     \
                       v_opacities)
    if (D == 4) SC_LAUNCH_BWD(4);
    else if (D == 3) SC_LAUNCH_BWD(3);
    else SC_LAUNCH_BWD(0);
#undef SC_LAUNCH_BWD
    SC_LAUNCH_CHECK();
    return SC_OK;
}
