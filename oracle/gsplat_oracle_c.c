/* CPU oracle in C -- TEST INFRASTRUCTURE ONLY (multi-core restatement of oracle/gsplat_oracle.py).
 *
 * Plain C99 + OpenMP restatement of the FORWARD of the gsplat operator path StreetCrafter calls at
 * street_gaussian/models/street_gaussian_renderer.py:219-280 (projection, tile intersection + stable
 * key sort, offset encode, spherical harmonics, alpha-composite), in the same normative op order as the
 * numpy oracle (every expression fp32, one rounding per operation: build with -ffp-contract=off, no
 * -ffast-math).  Integer outputs and the projection / SH floats are bit-identical to the numpy oracle
 * (tests/test_oracle_cpu.py pins one against the other); blended pixels differ from it only through
 * libm's expf (a few 1e-7).  Used as the checker at sizes where numpy is too slow and as bench.py's
 * `cpu_baseline` (kind "port", all host cores).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path never does.
 *
 * PARITY STATUS: parity unpinned for projection / intersection / rasterize (same reasons as
 * gsplat_oracle.py: the arithmetic lives in the un-pinned third-party CUDA package gsplat, absent from
 * /root/reference); the SH basis is pinned against street_gaussian/utils/sh_utils.py:57-112 through the
 * numpy oracle's fixture.  Algorithm: SURVEY.md Appendix A.1-A.5; version choices U1-U3 as in
 * gsplat_oracle.py (Jacobian clamp 1.3 tan(fov/2), radius floor 0.01).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ALPHA_MIN (1.0f / 255.0f)
#define ALPHA_MAX 0.999f
#define T_EPS 1e-4f

static inline float dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
    return (a0 * b0 + a1 * b1) + a2 * b2;
}

int sco_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void sco_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---- a1 fully_fused_projection (renderer.py:219-234; gsplat_oracle.py:97-190) ---- */
/* proj_clamp 0 = 1.3 tan(fov/2) on both sides, 1 = ((W - cx)/fx + 0.3 tan, cx/fx + 0.3 tan); radius_floor 0.01 or 0.1:
 * the two upstream-version-dependent constants (gsplat_oracle.py, PROJ_CLAMPS / RADIUS_FLOORS) */
void sco_projection_v(const float* means, const float* quats, const float* scales, const float* V /*4x4*/,
                      const float* K /*3x3*/, int64_t N, int width, int height, float eps2d, float near_plane,
                      float far_plane, float radius_clip, int proj_clamp, float radius_floor, int32_t* radii,
                      float* means2d, float* depths, float* conics, float* comps);
void sco_projection(const float* means, const float* quats, const float* scales, const float* V /*4x4*/,
                    const float* K /*3x3*/, int64_t N, int width, int height, float eps2d, float near_plane,
                    float far_plane, float radius_clip, int32_t* radii, float* means2d, float* depths,
                    float* conics, float* comps) {
    sco_projection_v(means, quats, scales, V, K, N, width, height, eps2d, near_plane, far_plane, radius_clip, 0, 0.01f, radii,
                     means2d, depths, conics, comps);
}
void sco_projection_v(const float* means, const float* quats, const float* scales, const float* V /*4x4*/,
                      const float* K /*3x3*/, int64_t N, int width, int height, float eps2d, float near_plane,
                      float far_plane, float radius_clip, int proj_clamp, float radius_floor, int32_t* radii,
                      float* means2d, float* depths, float* conics, float* comps) {
    const float W00 = V[0], W01 = V[1], W02 = V[2], tx_ = V[3];
    const float W10 = V[4], W11 = V[5], W12 = V[6], ty_ = V[7];
    const float W20 = V[8], W21 = V[9], W22 = V[10], tz_ = V[11];
    const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const float tanx = 0.5f * (float)width / fx, tany = 0.5f * (float)height / fy;
    const float Wf = (float)width, Hf = (float)height;
    float limxp, limxn, limyp, limyn;
    if (proj_clamp == 0) {
        limxp = limxn = 1.3f * tanx;
        limyp = limyn = 1.3f * tany;
    } else {
        limxp = (Wf - cx) / fx + 0.3f * tanx;
        limxn = cx / fx + 0.3f * tanx;
        limyp = (Hf - cy) / fy + 0.3f * tany;
        limyn = cy / fy + 0.3f * tany;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
        const float x = dot3(W00, mx, W01, my, W02, mz) + tx_;
        const float y = dot3(W10, mx, W11, my, W12, mz) + ty_;
        const float z = dot3(W20, mx, W21, my, W22, mz) + tz_;
        int valid = !((z < near_plane) || (z > far_plane));
        /* quaternion (wxyz) -> rotation, normalised in-op (general_utils.py:125-146) */
        float qw = quats[4 * i], qx = quats[4 * i + 1], qy = quats[4 * i + 2], qz = quats[4 * i + 3];
        const float n2 = ((qx * qx + qy * qy) + qz * qz) + qw * qw;
        const float inv = 1.0f / sqrtf(n2);
        qw = qw * inv; qx = qx * inv; qy = qy * inv; qz = qz * inv;
        const float x2 = qx * qx, y2 = qy * qy, z2 = qz * qz;
        const float xy = qx * qy, xz = qx * qz, yz = qy * qz;
        const float wx = qw * qx, wy = qw * qy, wz = qw * qz;
        const float R00 = 1.0f - 2.0f * (y2 + z2), R01 = 2.0f * (xy - wz), R02 = 2.0f * (xz + wy);
        const float R10 = 2.0f * (xy + wz), R11 = 1.0f - 2.0f * (x2 + z2), R12 = 2.0f * (yz - wx);
        const float R20 = 2.0f * (xz - wy), R21 = 2.0f * (yz + wx), R22 = 1.0f - 2.0f * (x2 + y2);
        const float s0 = scales[3 * i], s1 = scales[3 * i + 1], s2 = scales[3 * i + 2];
        const float M00 = R00 * s0, M01 = R01 * s1, M02 = R02 * s2;
        const float M10 = R10 * s0, M11 = R11 * s1, M12 = R12 * s2;
        const float M20 = R20 * s0, M21 = R21 * s1, M22 = R22 * s2;
        const float S00 = dot3(M00, M00, M01, M01, M02, M02), S01 = dot3(M00, M10, M01, M11, M02, M12);
        const float S02 = dot3(M00, M20, M01, M21, M02, M22), S11 = dot3(M10, M10, M11, M11, M12, M12);
        const float S12 = dot3(M10, M20, M11, M21, M12, M22), S22 = dot3(M20, M20, M21, M21, M22, M22);
        const float T00 = dot3(W00, S00, W01, S01, W02, S02), T01 = dot3(W00, S01, W01, S11, W02, S12);
        const float T02 = dot3(W00, S02, W01, S12, W02, S22), T10 = dot3(W10, S00, W11, S01, W12, S02);
        const float T11 = dot3(W10, S01, W11, S11, W12, S12), T12 = dot3(W10, S02, W11, S12, W12, S22);
        const float T20 = dot3(W20, S00, W21, S01, W22, S02), T21 = dot3(W20, S01, W21, S11, W22, S12);
        const float T22 = dot3(W20, S02, W21, S12, W22, S22);
        const float c00 = dot3(T00, W00, T01, W01, T02, W02), c01 = dot3(T00, W10, T01, W11, T02, W12);
        const float c02 = dot3(T00, W20, T01, W21, T02, W22), c11 = dot3(T10, W10, T11, W11, T12, W12);
        const float c12 = dot3(T10, W20, T11, W21, T12, W22), c22 = dot3(T20, W20, T21, W21, T22, W22);
        const float rz = 1.0f / z, rz2 = rz * rz;
        const float tx = z * fminf(limxp, fmaxf(-limxn, x * rz));
        const float ty = z * fminf(limyp, fmaxf(-limyn, y * rz));
        const float ja = fx * rz, jb = ((-fx) * tx) * rz2, jc = fy * rz, jd = ((-fy) * ty) * rz2;
        const float u0 = ja * c00 + jb * c02, u1 = ja * c01 + jb * c12, u2 = ja * c02 + jb * c22;
        const float v1 = jc * c11 + jd * c12, v2 = jc * c12 + jd * c22;
        const float a = u0 * ja + u2 * jb, b = u1 * jc + u2 * jd, c = v1 * jc + v2 * jd;
        const float m2x = (fx * x) * rz + cx, m2y = (fy * y) * rz + cy;
        const float det0 = a * c - b * b;
        const float a1 = a + eps2d, c1 = c + eps2d;
        const float det1 = a1 * c1 - b * b;
        /* np.maximum propagates NaN; fmaxf would not: spell it out */
        const float ratio = det0 / det1;
        const float comp = sqrtf(ratio != ratio ? ratio : (ratio > 0.0f ? ratio : 0.0f));
        valid &= !(det1 <= 0.0f);
        valid &= !(det1 != det1);
        const float con0 = c1 / det1, con1 = (-b) / det1, con2 = a1 / det1;
        const float bb = 0.5f * (a1 + c1);
        const float disc = bb * bb - det1;
        const float lam = bb + sqrtf(disc != disc ? disc : (disc > radius_floor ? disc : radius_floor));
        const float radius = ceilf(3.0f * sqrtf(lam));
        valid &= !(radius <= radius_clip);
        valid &= !(radius != radius);
        valid &= !((m2x + radius <= 0.0f) || (m2x - radius >= Wf) || (m2y + radius <= 0.0f) || (m2y - radius >= Hf));
        if (valid) {
            radii[i] = (int32_t)fminf(radius, 2147483520.0f);
            means2d[2 * i] = m2x; means2d[2 * i + 1] = m2y;
            depths[i] = z;
            conics[3 * i] = con0; conics[3 * i + 1] = con1; conics[3 * i + 2] = con2;
            comps[i] = comp;
        } else {
            radii[i] = 0;
            means2d[2 * i] = 0.0f; means2d[2 * i + 1] = 0.0f;
            depths[i] = 0.0f;
            conics[3 * i] = conics[3 * i + 1] = conics[3 * i + 2] = 0.0f;
            comps[i] = 0.0f;
        }
    }
}

/* ---- a3 isect_tiles (renderer.py:241-252; gsplat_oracle.py:200-252) ------------- */
static inline int tile_bits_for(int64_t n_tiles) {
    int b = 0;
    while (n_tiles > 0) { ++b; n_tiles >>= 1; }
    return b < 1 ? 1 : b;          /* floor(log2(n)) + 1 */
}

static inline void tile_rect(float mx, float my, int32_t radius, float ts, float tw, float th, int* x0, int* x1,
                             int* y0, int* y1) {
    if (radius <= 0) { *x0 = *x1 = *y0 = *y1 = 0; return; }
    const float tr = (float)radius / ts, tx = mx / ts, ty = my / ts;
    *x0 = (int)fmaxf(fminf(floorf(tx - tr), tw), 0.0f);
    *x1 = (int)fmaxf(fminf(ceilf(tx + tr), tw), 0.0f);
    *y0 = (int)fmaxf(fminf(floorf(ty - tr), th), 0.0f);
    *y1 = (int)fmaxf(fminf(ceilf(ty + tr), th), 0.0f);
}

/* pass 1: tiles_per_gauss; returns the total number of intersections */
int64_t sco_isect_count(const float* means2d, const int32_t* radii, int64_t CN, int tile_size, int tile_width,
                        int tile_height, int32_t* tiles_per_gauss) {
    int64_t total = 0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int64_t i = 0; i < CN; ++i) {
        int x0, x1, y0, y1;
        tile_rect(means2d[2 * i], means2d[2 * i + 1], radii[i], (float)tile_size, (float)tile_width,
                  (float)tile_height, &x0, &x1, &y0, &y1);
        const int c = (y1 - y0) * (x1 - x0);
        tiles_per_gauss[i] = c;
        total += c;
    }
    return total;
}

/* stable LSD radix sort of (key, value) pairs over key bits [0, end_bit), 8 bits per pass,
 * per-thread histograms over contiguous chunks (so the scatter keeps input order inside a digit) */
static void radix_sort_pairs(uint64_t* keys, int32_t* vals, uint64_t* tk, int32_t* tv, int64_t n, int end_bit) {
    int nthreads = sco_num_threads();
    if (n < 65536) nthreads = 1;
    int64_t* hist = (int64_t*)malloc(sizeof(int64_t) * 256 * (size_t)nthreads);
    uint64_t *src = keys, *dst = tk;
    int32_t *sv = vals, *dv = tv;
    int passes = 0;
    for (int shift = 0; shift < end_bit; shift += 8, ++passes) {
        memset(hist, 0, sizeof(int64_t) * 256 * (size_t)nthreads);
#pragma omp parallel num_threads(nthreads)
        {
#ifdef _OPENMP
            const int t = omp_get_thread_num();
#else
            const int t = 0;
#endif
            const int64_t beg = n * t / nthreads, end = n * (t + 1) / nthreads;
            int64_t* h = hist + 256 * (size_t)t;
            for (int64_t i = beg; i < end; ++i) ++h[(src[i] >> shift) & 255u];
#pragma omp barrier
#pragma omp single
            {
                int64_t run = 0;
                for (int d = 0; d < 256; ++d)
                    for (int q = 0; q < nthreads; ++q) {
                        const int64_t c = hist[256 * (size_t)q + d];
                        hist[256 * (size_t)q + d] = run;
                        run += c;
                    }
            }
            for (int64_t i = beg; i < end; ++i) {
                const int64_t p = h[(src[i] >> shift) & 255u]++;
                dst[p] = src[i];
                dv[p] = sv[i];
            }
        }
        uint64_t* tmpk = src; src = dst; dst = tmpk;
        int32_t* tmpv = sv; sv = dv; dv = tmpv;
    }
    if (passes & 1) {
        memcpy(keys, src, sizeof(uint64_t) * (size_t)n);
        memcpy(vals, sv, sizeof(int32_t) * (size_t)n);
    }
    free(hist);
}

/* pass 2: emit (gaussian-major, row-major over the rectangle), stable sort by key, offsets (A.3) */
int sco_isect_emit_sort(const float* means2d, const int32_t* radii, const float* depths, int C, int64_t N,
                        int tile_size, int tile_width, int tile_height, const int32_t* tiles_per_gauss,
                        int64_t total, int sort, int64_t* isect_ids, int32_t* flatten_ids,
                        int32_t* isect_offsets /* [C*th*tw], nullable */) {
    const int64_t CN = (int64_t)C * N;
    const int64_t n_tiles = (int64_t)tile_width * tile_height;
    const int tb = tile_bits_for(n_tiles);
    int cb = 0;
    for (int c = C; c > 0; c >>= 1) ++cb;
    if (cb < 1) cb = 1;
    int64_t* starts = (int64_t*)malloc(sizeof(int64_t) * (size_t)(CN + 1));
    if (!starts) return -1;
    int64_t run = 0;
    for (int64_t i = 0; i < CN; ++i) { starts[i] = run; run += tiles_per_gauss[i]; }
    starts[CN] = run;
    if (run != total) { free(starts); return -2; }
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t i = 0; i < CN; ++i) {
        if (tiles_per_gauss[i] <= 0) continue;
        int x0, x1, y0, y1;
        tile_rect(means2d[2 * i], means2d[2 * i + 1], radii[i], (float)tile_size, (float)tile_width,
                  (float)tile_height, &x0, &x1, &y0, &y1);
        const int64_t cid = i / N;
        uint32_t dbits;
        memcpy(&dbits, &depths[i], 4);
        int64_t p = starts[i];
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = x0; tx < x1; ++tx) {
                isect_ids[p] = (cid << (32 + tb)) | (((int64_t)ty * tile_width + tx) << 32) | (int64_t)dbits;
                flatten_ids[p] = (int32_t)i;
                ++p;
            }
    }
    free(starts);
    if (sort && total > 0) {
        uint64_t* tk = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)total);
        int32_t* tv = (int32_t*)malloc(sizeof(int32_t) * (size_t)total);
        if (!tk || !tv) { free(tk); free(tv); return -1; }
        radix_sort_pairs((uint64_t*)isect_ids, flatten_ids, tk, tv, total, 32 + tb + cb);
        free(tk); free(tv);
    }
    if (isect_offsets) {
        /* lower bound of every (camera, tile) among the sorted keys; tiles after the last key -> total */
        const int64_t nb = (int64_t)C * n_tiles;
#pragma omp parallel for schedule(static)
        for (int64_t q = 0; q < nb; ++q) {
            const int64_t cam = q / n_tiles, tile = q - cam * n_tiles;
            const int64_t want = (cam << tb) | tile;      /* == key >> 32 */
            int64_t lo = 0, hi = total;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if ((isect_ids[mid] >> 32) < want) lo = mid + 1; else hi = mid;
            }
            isect_offsets[q] = (int32_t)lo;
        }
    }
    return 0;
}

/* ---- a6 spherical_harmonics (renderer.py:259; gsplat_oracle.py:270-340) ---------- */
static void sh_basis(int degree, float x, float y, float z, float* Y) {
    Y[0] = 0.2820947917738781f;
    if (degree < 1) return;
    const float inorm = 1.0f / sqrtf((x * x + y * y) + z * z);
    x = x * inorm; y = y * inorm; z = z * inorm;
    const float c1 = 0.48860251190292f;
    Y[1] = (-c1) * y; Y[2] = c1 * z; Y[3] = (-c1) * x;
    if (degree < 2) return;
    const float z2 = z * z;
    const float fTmp0B = -1.092548430592079f * z;
    const float fC1 = x * x - y * y;
    const float fS1 = 2.0f * x * y;
    const float pSH6 = 0.9461746957575601f * z2 - 0.3153915652525201f;
    Y[4] = 0.5462742152960395f * fS1; Y[5] = fTmp0B * y; Y[6] = pSH6; Y[7] = fTmp0B * x;
    Y[8] = 0.5462742152960395f * fC1;
    if (degree < 3) return;
    const float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
    const float fTmp1B = 1.445305721320277f * z;
    const float fC2 = x * fC1 - y * fS1;
    const float fS2 = x * fS1 + y * fC1;
    const float pSH12 = z * (1.865881662950577f * z2 - 1.119528997770346f);
    Y[9] = -0.5900435899266435f * fS2; Y[10] = fTmp1B * fS1; Y[11] = fTmp0C * y; Y[12] = pSH12;
    Y[13] = fTmp0C * x; Y[14] = fTmp1B * fC1; Y[15] = -0.5900435899266435f * fC2;
    if (degree < 4) return;
    const float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    const float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f;
    const float fTmp2B = -1.770130769779931f * z;
    const float fC3 = x * fC2 - y * fS2;
    const float fS3 = x * fS2 + y * fC2;
    Y[16] = 0.6258357354491763f * fS3; Y[17] = fTmp2B * fS2; Y[18] = fTmp1C * fS1; Y[19] = fTmp0D * y;
    Y[20] = 1.984313483298443f * z * pSH12 + -1.006230589874905f * pSH6;
    Y[21] = fTmp0D * x; Y[22] = fTmp1C * fC1; Y[23] = fTmp2B * fC2; Y[24] = 0.6258357354491763f * fC3;
}

/* dirs [M,3] (not normalised), coeffs [M,K,3], masks u8[M] nullable -> colors [M,3] */
void sco_spherical_harmonics(int degree, const float* dirs, const float* coeffs, const uint8_t* masks, int64_t M,
                             int K, float* colors) {
    const int nb = (degree + 1) * (degree + 1);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; ++i) {
        if (masks && !masks[i]) { colors[3 * i] = colors[3 * i + 1] = colors[3 * i + 2] = 0.0f; continue; }
        float Y[25];
        sh_basis(degree, dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], Y);
        const float* c = coeffs + (size_t)i * K * 3;
        for (int ch = 0; ch < 3; ++ch) {
            float out = Y[0] * c[ch];
            for (int k = 1; k < nb; ++k) out = out + Y[k] * c[3 * k + ch];
            colors[3 * i + ch] = out;
        }
    }
}

/* ---- a9 rasterize_to_pixels forward (renderer.py:267-280; gsplat_oracle.py:344-444) */
void sco_rasterize_cond(const float* means2d, const float* conics, const float* colors, const float* opacities,
                   const float* backgrounds /* [C,D] nullable */, int C, int64_t N, int D, int width, int height,
                   int tile_size, int tile_width, int tile_height, const int32_t* isect_offsets,
                   const int32_t* flatten_ids, int64_t n_isects, float* render_colors, float* render_alphas,
                   int32_t* last_ids /* nullable */, uint8_t* unstable /* [C,H,W] nullable */, float unstable_rel,
                   float unstable_cond, float* cond_bound /* [C,H,W] nullable; needs unstable and unstable_cond > 0 */) {
    /* `unstable`: pixels where some alpha / sigma / transmittance the walk looked at sits within
     * `unstable_rel` of a hard threshold, so that a 1-ulp change of exp() may legitimately flip the skip /
     * terminate decision (same rule as gsplat_oracle.rasterize_to_pixels(return_unstable=True)).
     * `unstable_cond` > 0 widens the windows by the rounding-error bound of the quantities themselves:
     * sigma is a sum of terms of magnitude S = (|A| dx^2 + |C| dy^2) / 2 + |B dx dy|, so another evaluation
     * order moves it by about unstable_cond * 2^-24 * S (a big rotated splat has S >> sigma), alpha by that
     * relative amount, and the transmittance by the accumulated relative errors of its factors.
     * The byte written is a code: bit 0 = flagged under the fixed windows alone (what unstable_cond = 0 flags),
     * bit 1 = flagged at all; non-zero == unstable.
     * `cond_bound` receives, per pixel, the first-order bound of what that same rounding error does to the blend
     * itself (no threshold involved): sum_i vis_i * (ea_i + 3 eps + terr_i) + T * terr, with ea_i the bound of
     * alpha_i's relative error and terr_i that of the transmittance up to and including splat i -- in units of the colour
     * scale.  Giant splats seen from close by (sigma's terms in the hundreds, sigma itself ~ 1) reach 1e-4 with it:
     * there fp32 itself, not the implementation, limits the agreement of two evaluation orders. */
    (void)N;
    const float EPS24 = 0x1p-24f;
    const float ce = unstable_cond > 0.0f ? unstable_cond * EPS24 : 0.0f;
    const int64_t n_tiles = (int64_t)C * tile_width * tile_height;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t tflat = 0; tflat < n_tiles; ++tflat) {
        const int cam = (int)(tflat / ((int64_t)tile_width * tile_height));
        const int tid = (int)(tflat - (int64_t)cam * tile_width * tile_height);
        const int ty = tid / tile_width, tx = tid - ty * tile_width;
        const int64_t start = isect_offsets[tflat];
        const int64_t end = (tflat + 1 < n_tiles) ? isect_offsets[tflat + 1] : n_isects;
        const int y0 = ty * tile_size, x0 = tx * tile_size;
        const int y1 = y0 + tile_size < height ? y0 + tile_size : height;
        const int x1 = x0 + tile_size < width ? x0 + tile_size : width;
        float acc[32];
        for (int py = y0; py < y1; ++py)
            for (int px = x0; px < x1; ++px) {
                const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
                float T = 1.0f;
                int32_t cur = 0;
                int unst = 0;
                float terr = 0.0f;          /* accumulated relative error bound of T (unstable_cond > 0 only) */
                float bsum = 0.0f;          /* cond_bound's sum */
                for (int d = 0; d < D; ++d) acc[d] = 0.0f;
                for (int64_t k = start; k < end; ++k) {
                    const int64_t g = flatten_ids[k];
                    const float dx = means2d[2 * g] - pxf, dy = means2d[2 * g + 1] - pyf;
                    const float ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
                    const float sigma = 0.5f * ((ca * dx) * dx + (cc * dy) * dy) + (cb * dx) * dy;
                    const float e = opacities[g] * expf(-sigma);
                    const float alpha = e != e ? e : (e < ALPHA_MAX ? e : ALPHA_MAX);
                    float wa = unstable_rel, ws = 1e-6f, ea = 0.0f;
                    if (unstable && ce > 0.0f) {
                        const float S = 0.5f * (fabsf((ca * dx) * dx) + fabsf((cc * dy) * dy)) + fabsf((cb * dx) * dy);
                        ea = ce * S;                 /* error bound of sigma == relative error bound of alpha */
                        wa = unstable_rel + ea;
                        ws = 1e-6f + ea;
                    }
                    if (unstable && ((fabsf(alpha - ALPHA_MIN) <= wa * ALPHA_MIN) || (fabsf(sigma) <= ws))) {
                        unst |= 2;
                        if ((fabsf(alpha - ALPHA_MIN) <= unstable_rel * ALPHA_MIN) || (fabsf(sigma) <= 1e-6f)) unst |= 1;
                    }
                    if ((sigma < 0.0f) || (alpha < ALPHA_MIN) || (alpha != alpha)) continue;
                    const float Tn = T * (1.0f - alpha);
                    float wt = unstable_rel * 10.0f;
                    if (unstable && ce > 0.0f) {
                        terr = terr + ((alpha * (ea + 3.0f * EPS24)) / (1.0f - alpha) + EPS24);
                        wt = unstable_rel * 10.0f + terr;
                    }
                    if (unstable && (fabsf(Tn - T_EPS) <= wt * T_EPS)) {
                        unst |= 2;
                        if (fabsf(Tn - T_EPS) <= (unstable_rel * 10.0f) * T_EPS) unst |= 1;
                    }
                    if (Tn <= T_EPS) break;
                    const float vis = alpha * T;
                    if (unstable && ce > 0.0f) bsum = bsum + vis * ((ea + 3.0f * EPS24) + terr);
                    const float* c = colors + (size_t)g * D;
                    for (int d = 0; d < D; ++d) acc[d] = acc[d] + c[d] * vis;
                    cur = (int32_t)k;
                    T = Tn;
                }
                const int64_t pix = ((int64_t)cam * height + py) * width + px;
                for (int d = 0; d < D; ++d)
                    render_colors[pix * D + d] = backgrounds ? acc[d] + T * backgrounds[cam * D + d] : acc[d];
                render_alphas[pix] = 1.0f - T;
                if (last_ids) last_ids[pix] = cur;
                if (unstable) unstable[pix] = (uint8_t)unst;
                if (cond_bound) cond_bound[pix] = bsum + T * terr;
            }
    }
}

void sco_rasterize(const float* means2d, const float* conics, const float* colors, const float* opacities,
                   const float* backgrounds, int C, int64_t N, int D, int width, int height, int tile_size,
                   int tile_width, int tile_height, const int32_t* isect_offsets, const int32_t* flatten_ids,
                   int64_t n_isects, float* render_colors, float* render_alphas, int32_t* last_ids, uint8_t* unstable,
                   float unstable_rel) {
    sco_rasterize_cond(means2d, conics, colors, opacities, backgrounds, C, N, D, width, height, tile_size, tile_width,
                       tile_height, isect_offsets, flatten_ids, n_isects, render_colors, render_alphas, last_ids,
                       unstable, unstable_rel, 0.0f, NULL);
}
