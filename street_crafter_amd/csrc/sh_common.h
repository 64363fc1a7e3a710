// Real SH basis (street_gaussian/utils/sh_utils.py:57-112 on dir/|dir|) and the
// colour evaluation, shared by sh_fwd_kernel (sh.hip) and the fused projection + SH kernel
// (fused_fwd.hip).  No FMA contraction, oracle op order: bit-identical to oracle/gsplat_oracle.py.
#pragma once
#include "sc_common.h"

#pragma clang fp contract(off)

namespace {

// Real SH basis up to degree 4 on the normalised direction; Y[0..(deg+1)^2).
template <int DEG>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float* Y) {
    Y[0] = 0.2820947917738781f;
    if (DEG < 1) return;
    const float inorm = 1.0f / sqrtf((x * x + y * y) + z * z);
    x *= inorm; y *= inorm; z *= inorm;
    const float c1 = 0.48860251190292f;
    Y[1] = (-c1) * y; Y[2] = c1 * z; Y[3] = (-c1) * x;
    if (DEG < 2) return;
    const float z2 = z * z;
    const float fTmp0B = -1.092548430592079f * z;
    const float fC1 = x * x - y * y;
    const float fS1 = 2.0f * x * y;
    const float pSH6 = 0.9461746957575601f * z2 - 0.3153915652525201f;
    Y[4] = 0.5462742152960395f * fS1;
    Y[5] = fTmp0B * y;
    Y[6] = pSH6;
    Y[7] = fTmp0B * x;
    Y[8] = 0.5462742152960395f * fC1;
    if (DEG < 3) return;
    const float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
    const float fTmp1B = 1.445305721320277f * z;
    const float fC2 = x * fC1 - y * fS1;
    const float fS2 = x * fS1 + y * fC1;
    const float pSH12 = z * (1.865881662950577f * z2 - 1.119528997770346f);
    Y[9] = -0.5900435899266435f * fS2;
    Y[10] = fTmp1B * fS1;
    Y[11] = fTmp0C * y;
    Y[12] = pSH12;
    Y[13] = fTmp0C * x;
    Y[14] = fTmp1B * fC1;
    Y[15] = -0.5900435899266435f * fC2;
    if (DEG < 4) return;
    const float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    const float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f;
    const float fTmp2B = -1.770130769779931f * z;
    const float fC3 = x * fC2 - y * fS2;
    const float fS3 = x * fS2 + y * fC2;
    Y[16] = 0.6258357354491763f * fS3;
    Y[17] = fTmp2B * fS2;
    Y[18] = fTmp1C * fS1;
    Y[19] = fTmp0D * y;
    Y[20] = 1.984313483298443f * z * pSH12 + -1.006230589874905f * pSH6;
    Y[21] = fTmp0D * x;
    Y[22] = fTmp1C * fC1;
    Y[23] = fTmp2B * fC2;
    Y[24] = 0.6258357354491763f * fC3;
}

// colour = sum_k Y_k(dir) * coeffs[k][:], coeffs laid out [K][3]
template <int DEG>
__device__ __forceinline__ void sh_eval(float dx, float dy, float dz, const float* __restrict__ c, float& r,
                                        float& g, float& b) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    float Y[NB];
    sh_basis<DEG>(dx, dy, dz, Y);
    r = Y[0] * c[0]; g = Y[0] * c[1]; b = Y[0] * c[2];
#pragma unroll
    for (int k = 1; k < NB; ++k) {
        r = r + Y[k] * c[k * 3 + 0];
        g = g + Y[k] * c[k * 3 + 1];
        b = b + Y[k] * c[k * 3 + 2];
    }
}

}  // namespace

#pragma clang fp contract(fast)
