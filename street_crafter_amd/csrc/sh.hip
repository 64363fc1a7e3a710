// a6: spherical_harmonics forward / backward for gfx950.
// Replaces gsplat.rendering.spherical_harmonics as called at
// street_gaussian/models/street_gaussian_renderer.py:259 (semantics: SURVEY.md A.4; the basis is
// the one in the reference's street_gaussian/utils/sh_utils.py:57-112, evaluated on dir/|dir|).
// HBM-bound streaming kernel, one lane per row.  Compiled without FMA contraction and in the
// oracle's op order, so the forward is bit-identical to oracle/gsplat_oracle.py.
#include "sh_common.h"

#pragma clang fp contract(off)

namespace {

template <int DEG>
__global__ __launch_bounds__(256) void sh_fwd_kernel(const float* __restrict__ dirs,
                                                     const float* __restrict__ coeffs,
                                                     const uint8_t* __restrict__ masks, int64_t M,
                                                     int K, float* __restrict__ colors) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    float r = 0.f, g = 0.f, b = 0.f;
    if (!masks || masks[i])
        sh_eval<DEG>(dirs[i * 3 + 0], dirs[i * 3 + 1], dirs[i * 3 + 2], coeffs + i * (int64_t)K * 3, r, g, b);
    colors[i * 3 + 0] = r; colors[i * 3 + 1] = g; colors[i * 3 + 2] = b;
}

}  // namespace

#pragma clang fp contract(fast)

namespace {

// d(basis)/d(unit dir) contracted with w_k = sum_ch v_color[ch]*coeff[k][ch]; returns dL/d(unit dir).
template <int DEG>
__device__ __forceinline__ void sh_basis_vjp(float x, float y, float z, const float* w, float* vd) {
    float vx = 0.f, vy = 0.f, vz = 0.f;
    if (DEG >= 1) {
        const float c1 = 0.48860251190292f;
        vy += -c1 * w[1]; vz += c1 * w[2]; vx += -c1 * w[3];
    }
    if (DEG >= 2) {
        const float z2 = z * z;
        const float fTmp0B = -1.092548430592079f * z;
        const float fC1 = x * x - y * y, fS1 = 2.f * x * y;
        // derivatives of helpers
        const float fTmp0B_z = -1.092548430592079f;
        const float fC1_x = 2.f * x, fC1_y = -2.f * y;
        const float fS1_x = 2.f * y, fS1_y = 2.f * x;
        const float pSH6_z = 2.f * 0.9461746957575601f * z;
        const float k4 = 0.5462742152960395f;
        vx += k4 * fS1_x * w[4] + fTmp0B * w[7] + k4 * fC1_x * w[8];
        vy += k4 * fS1_y * w[4] + fTmp0B * w[5] + k4 * fC1_y * w[8];
        vz += fTmp0B_z * y * w[5] + pSH6_z * w[6] + fTmp0B_z * x * w[7];
        if (DEG >= 3) {
            const float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
            const float fTmp1B = 1.445305721320277f * z;
            const float fC2 = x * fC1 - y * fS1, fS2 = x * fS1 + y * fC1;
            const float fTmp0C_z = -2.285228997322329f * 2.f * z;
            const float fTmp1B_z = 1.445305721320277f;
            const float fC2_x = fC1 + x * fC1_x - y * fS1_x, fC2_y = x * fC1_y - fS1 - y * fS1_y;
            const float fS2_x = fS1 + x * fS1_x + y * fC1_x, fS2_y = x * fS1_y + fC1 + y * fC1_y;
            const float pSH12 = z * (1.865881662950577f * z2 - 1.119528997770346f);
            const float pSH12_z = 3.f * 1.865881662950577f * z2 - 1.119528997770346f;
            const float k9 = -0.5900435899266435f;
            vx += k9 * fS2_x * w[9] + fTmp1B * fS1_x * w[10] + fTmp0C * w[13] + fTmp1B * fC1_x * w[14] + k9 * fC2_x * w[15];
            vy += k9 * fS2_y * w[9] + fTmp1B * fS1_y * w[10] + fTmp0C * w[11] + fTmp1B * fC1_y * w[14] + k9 * fC2_y * w[15];
            vz += fTmp1B_z * fS1 * w[10] + fTmp0C_z * y * w[11] + pSH12_z * w[12] + fTmp0C_z * x * w[13] + fTmp1B_z * fC1 * w[14];
            if (DEG >= 4) {
                const float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
                const float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f;
                const float fTmp2B = -1.770130769779931f * z;
                const float fC3_x = fC2 + x * fC2_x - y * fS2_x, fC3_y = x * fC2_y - fS2 - y * fS2_y;
                const float fS3_x = fS2 + x * fS2_x + y * fC2_x, fS3_y = x * fS2_y + fC2 + y * fC2_y;
                const float fTmp0D_z = 3.f * -4.683325804901025f * z2 + 2.007139630671868f;
                const float fTmp1C_z = 2.f * 3.31161143515146f * z;
                const float fTmp2B_z = -1.770130769779931f;
                const float pSH6 = 0.9461746957575601f * z2 - 0.3153915652525201f;
                const float pSH20_z = 1.984313483298443f * (pSH12 + z * pSH12_z) + -1.006230589874905f * pSH6_z;
                (void)pSH6;
                const float k16 = 0.6258357354491763f;
                vx += k16 * fS3_x * w[16] + fTmp2B * fS2_x * w[17] + fTmp1C * fS1_x * w[18] + fTmp0D * w[21] + fTmp1C * fC1_x * w[22] + fTmp2B * fC2_x * w[23] + k16 * fC3_x * w[24];
                vy += k16 * fS3_y * w[16] + fTmp2B * fS2_y * w[17] + fTmp1C * fS1_y * w[18] + fTmp0D * w[19] + fTmp1C * fC1_y * w[22] + fTmp2B * fC2_y * w[23] + k16 * fC3_y * w[24];
                vz += fTmp2B_z * fS2 * w[17] + fTmp1C_z * fS1 * w[18] + fTmp0D_z * y * w[19] + pSH20_z * w[20] + fTmp0D_z * x * w[21] + fTmp1C_z * fC1 * w[22] + fTmp2B_z * fC2 * w[23];
            }
        }
    }
    vd[0] = vx; vd[1] = vy; vd[2] = vz;
}

// VEC: the rows of coeffs / v_coeffs are a multiple of 16 B (K * 3 floats, K = 4 or 16 in the reference's scenes):
// they are read and written as float4 -- 12 scalar stores of one lane's 48-B row touch the same cache lines 12
// times over (40 -> 30 us at 1 M Gaussians, degree 1).
template <int DEG, bool VEC>
__global__ __launch_bounds__(256) void sh_bwd_kernel(const float* __restrict__ dirs,
                                                     const float* __restrict__ coeffs,
                                                     const uint8_t* __restrict__ masks, int64_t M,
                                                     int K, const float* __restrict__ v_colors,
                                                     float* __restrict__ v_coeffs,
                                                     float* __restrict__ v_dirs) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    constexpr int NV = (NB * 3 + 3) / 4;           // float4 chunks that hold the NB live coefficients
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    float* vc = v_coeffs + i * (int64_t)K * 3;
    const bool on = !masks || masks[i];
    float vdx = 0.f, vdy = 0.f, vdz = 0.f;
    if (on) {
        const float dx = dirs[i * 3 + 0], dy = dirs[i * 3 + 1], dz = dirs[i * 3 + 2];
        float Y[NB];
        sh_basis<DEG>(dx, dy, dz, Y);
        const float vr = v_colors[i * 3 + 0], vg = v_colors[i * 3 + 1], vb = v_colors[i * 3 + 2];
        if (VEC) {
            float o[NV * 4];
#pragma unroll
            for (int f = 0; f < NV * 4; ++f) o[f] = (f < NB * 3) ? Y[f / 3] * ((f % 3) == 0 ? vr : ((f % 3) == 1 ? vg : vb)) : 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j)
                reinterpret_cast<float4*>(vc)[j] = make_float4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
            for (int j = NV; j < (K * 3) / 4; ++j) reinterpret_cast<float4*>(vc)[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                vc[k * 3 + 0] = Y[k] * vr; vc[k * 3 + 1] = Y[k] * vg; vc[k * 3 + 2] = Y[k] * vb;
            }
            for (int k = NB; k < K; ++k) { vc[k * 3 + 0] = 0.f; vc[k * 3 + 1] = 0.f; vc[k * 3 + 2] = 0.f; }
        }
        if (v_dirs && DEG >= 1) {
            const float* c = coeffs + i * (int64_t)K * 3;
            float cf[VEC ? NV * 4 : 1];
            if (VEC) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const float4 q = reinterpret_cast<const float4*>(c)[j];
                    cf[4 * j] = q.x; cf[4 * j + 1] = q.y; cf[4 * j + 2] = q.z; cf[4 * j + 3] = q.w;
                }
            }
            float w[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k)
                w[k] = VEC ? vr * cf[k * 3 + 0] + vg * cf[k * 3 + 1] + vb * cf[k * 3 + 2]
                           : vr * c[k * 3 + 0] + vg * c[k * 3 + 1] + vb * c[k * 3 + 2];
            const float inorm = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
            const float ux = dx * inorm, uy = dy * inorm, uz = dz * inorm;
            float vu[3];
            sh_basis_vjp<DEG>(ux, uy, uz, w, vu);
            // through u = d/|d|:  v_d = (v_u - (v_u . u) u) / |d|
            const float dp = vu[0] * ux + vu[1] * uy + vu[2] * uz;
            vdx = (vu[0] - dp * ux) * inorm;
            vdy = (vu[1] - dp * uy) * inorm;
            vdz = (vu[2] - dp * uz) * inorm;
        }
    } else if (VEC) {
        for (int j = 0; j < (K * 3) / 4; ++j) reinterpret_cast<float4*>(vc)[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (int k = 0; k < K; ++k) { vc[k * 3 + 0] = 0.f; vc[k * 3 + 1] = 0.f; vc[k * 3 + 2] = 0.f; }
    }
    if (v_dirs) { v_dirs[i * 3 + 0] = vdx; v_dirs[i * 3 + 1] = vdy; v_dirs[i * 3 + 2] = vdz; }
}

}  // namespace

extern "C" int sc_sh_fwd(int degree, const float* dirs, const float* coeffs, const uint8_t* masks,
                         int64_t M, int K, float* colors, sc_stream_t stream) {
    if (degree < 0 || degree > 4 || M < 0 || K < (degree + 1) * (degree + 1)) return SC_EINVAL;
    if (M == 0) return SC_OK;
    if (!dirs || !coeffs || !colors) return SC_EINVAL;
    const int64_t nb = (M + 255) / 256;
    if (nb > 0x7fffffff) return SC_EINVAL;
    dim3 grid((unsigned)nb), block(256);
    switch (degree) {
        case 0: hipLaunchKernelGGL(sh_fwd_kernel<0>, grid, block, 0, sc_s(stream), dirs, coeffs, masks, M, K, colors); break;
        case 1: hipLaunchKernelGGL(sh_fwd_kernel<1>, grid, block, 0, sc_s(stream), dirs, coeffs, masks, M, K, colors); break;
        case 2: hipLaunchKernelGGL(sh_fwd_kernel<2>, grid, block, 0, sc_s(stream), dirs, coeffs, masks, M, K, colors); break;
        case 3: hipLaunchKernelGGL(sh_fwd_kernel<3>, grid, block, 0, sc_s(stream), dirs, coeffs, masks, M, K, colors); break;
        default: hipLaunchKernelGGL(sh_fwd_kernel<4>, grid, block, 0, sc_s(stream), dirs, coeffs, masks, M, K, colors); break;
    }
    SC_LAUNCH_CHECK();
    return SC_OK;
}

extern "C" int sc_sh_bwd(int degree, const float* dirs, const float* coeffs, const uint8_t* masks,
                         int64_t M, int K, const float* v_colors, float* v_coeffs, float* v_dirs,
                         sc_stream_t stream) {
    if (degree < 0 || degree > 4 || M < 0 || K < (degree + 1) * (degree + 1)) return SC_EINVAL;
    if (M == 0) return SC_OK;
    if (!dirs || !coeffs || !v_colors || !v_coeffs) return SC_EINVAL;
    const int64_t nb = (M + 255) / 256;
    if (nb > 0x7fffffff) return SC_EINVAL;
    dim3 grid((unsigned)nb), block(256);
    // float4 rows: K * 3 floats a multiple of 4 and both arrays 16-B aligned (torch allocations are)
    const bool vec = (K * 3) % 4 == 0 && ((uintptr_t)coeffs % 16) == 0 && ((uintptr_t)v_coeffs % 16) == 0;
#define SC_SH_BWD(D)                                                                                              \
    if (vec) hipLaunchKernelGGL((sh_bwd_kernel<D, true>), grid, block, 0, sc_s(stream), dirs, coeffs, masks, M, K, \
                                v_colors, v_coeffs, v_dirs);                                                       \
    else hipLaunchKernelGGL((sh_bwd_kernel<D, false>), grid, block, 0, sc_s(stream), dirs, coeffs, masks, M, K,     \
                            v_colors, v_coeffs, v_dirs)
    switch (degree) {
        case 0: SC_SH_BWD(0); break;
        case 1: SC_SH_BWD(1); break;
        case 2: SC_SH_BWD(2); break;
        case 3: SC_SH_BWD(3); break;
        default: SC_SH_BWD(4); break;
    }
#undef SC_SH_BWD
    SC_LAUNCH_CHECK();
    return SC_OK;
}
